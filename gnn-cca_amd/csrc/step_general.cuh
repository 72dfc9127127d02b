#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// One message-passing step (MetaLayer.forward, models/mpn.py:32-54) fused with the edge encoder on step 1
// (mpn.py:137) and the edge classifier on classifying steps (mpn.py:290-293).
//
// Work split: a source node's edge segment is owned by `wps` (1, 2 or 4) waves of one workgroup; a wave walks
// its share in chunks of 64 edges (lane = edge).  Per chunk:
//   VALU : e' = ReLU(P_src[node] + P_dst[col] + W_ee e)           6 x (2 + 6|12) FMAs per edge
//          classifier logit (6 -> C1 -> 1) on classifying steps
//   MFMA : two 32-edge tiles, D[edge][channel] = Q[node][channel] + sum_k e'[edge][k] Wne[channel][k] as three
//          v_mfma_f32_32x32x2_f32 each (K = 6 exactly, 32 channels = one tile: no padding waste).  The A operand
//          (edge-major) comes straight from the VALU registers through one v_permlane32_swap per feature pair.
//   the accumulator layout puts the CHANNEL on the lane and the 32 edges of a tile in registers/half-waves, so
//   the per-source reduction is 16 in-register adds + one cross-half add: no atomics, no LDS, fixed order.
// After its segment a wave group reduces across its waves through LDS and projects h' for the next step.
// ------------------------------------------------------------------------------------------------------------
struct StepParams {
    const float* blob;
    const int* seg_ptr;
    const int* col32;
    const int* perm;
    unsigned* flags;         // [0] graph flags (plan), [1] column-range verdict (step 1)
    const float* edge_attr;
    float* e;
    float* e0;
    const float* pd_in;
    const float* psq_in;
    float* pd_out;
    float* psq_out;
    const float* h0;
    float* trace_h;
    float* trace_e;
    float* trace_e_enc;
    float* logits;
    float* logits_in;        // mpn_step_pipe_kernel<..., CIN>: where the logits of the PREVIOUS step go (classified from this step's input state)
    long long e_stride;
    int off_wee, off_wneb, off_projwT, off_projb, off_encw, off_encb, off_cw1, off_cb1, off_cw2, off_cb2;
    int off_fast;
    int off_wnebf;   // BlobHeader::wne_bf16
    int cls_layers, cls_hidden;  // cls_layers == 0: this step does not classify
    int N, E, edge_in, attr_vec, first, update, agg, reatt_n, wps, store_e, hin, pd_lds, stamp_slot, e_bf16;
    // padded edge-state layout of big, nearly regular batches: node i owns the slots [i * ell_S, (i + 1) * ell_S) of every
    // feature plane (ell_S a multiple of 32 floats = one 128-B line), so no line is shared by two segments; 0: compact
    // CSR order.  Chosen by the host from E/N alone; the plan raises GNNCCA_GRAPH_IRREGULAR when a degree exceeds it and
    // the kernels then use the compact order for this forward.
    int ell_S;
    int nt_store, nt_load;   // non-temporal policy of the specialised step kernel's streams (see step_fast.cuh)
    DropCfg drop;            // train-mode Dropout (general kernel only); all p == 0 in eval
    int step_no, cls_no;     // 1-based step, 0-based index of this classified step: the dropout streams
    int msg_f32;             // 1: the node message on f32-input MFMAs (an ordered fp32 FMA chain), the arithmetic of rounds 1-2: graphs of
                             // <= 512 nodes, where a launch is a chain of dependent latencies and the split-bf16 form's operand
                             // construction costs more than its matrix-pipe overlap returns (1 x dense256: 4.6 vs 4.8 us per step);
                             // 0: split-bf16 (msg_bf16.cuh).  One rule for the traced and the fast kernels: same bits either way
    // Column ranges (round 4).  Every graph the reference builds (inference.py:209-216: per camera, cartesian_prod of its detections with
    // every detection of the other cameras, targets ascending) and every dense graph gives a source node at most TWO contiguous runs of
    // target ids.  Step 1 of the specialised kernels reads the target ids anyway and derives, per node, (start1, len1, start2 - len1,
    // number of breaks) into `rng`; a node with more than one break raises flags[1].  Steps 2 ... L then COMPUTE the target id of
    // position q of a segment -- q + (q < len1 ? start1 : start2 - len1) -- instead of streaming col32 (4 of 56 B per edge) and, more
    // to the point, request the P_dst gather together with the edge state instead of one dependent round trip later.  flags[1] != 0
    // (graphs with arbitrary columns): every step streams col32 as before.  nullptr: off -- the default (GNNCCA_OPT_COLUMN_RANGES
    // asks for it): measured, the id load and the gather behind it are not on the launches' critical path (include/gnncca_mpn.h).
    int* rng;
    // mpn_step_pipe_kernel addresses the workspace through ONE buffer descriptor: its base, size and the byte offsets of the regions
    const void* ws_base;
    unsigned long long ws_bytes;
    unsigned so_e, so_col, so_perm, so_pd;
    int npw;                 // nodes per wave of mpn_step_pipe_kernel's message steps: 1, or 2 (step_pipe.cuh: NPW)
    int diag;                // GNNCCA_DIAG experiments (0 in production): bit 0 = timing-only run of mpn_step_pipe_kernel with
                             // zero-record stream descriptors (no HBM traffic: what the arithmetic alone costs)
};

template <bool REATT_E, bool MSG, bool AGG_MAX>
__global__ __launch_bounds__(256) void mpn_step_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int EFIN = REATT_E ? 2 * kEF : kEF;
    float* s_proj = smem;                                   // [hin][48]   (MSG)
    float* s_part = smem + (MSG ? p.hin * kProjOut : 0);    // [4][32]
    float* s_pd = s_part + 4 * kH;                          // [N][8]      (pd_lds: small graphs keep P_dst on chip)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    // Prologue: every load that does not depend on another one is issued before the first wait (flag word, CSR
    // offsets, the node's P_src/Q row, the MFMA B operand, the projection weights for the epilogue).
    const unsigned gflags = p.flags[0];
    const int wps = p.wps;
    const int node = blockIdx.x * (4 / wps) + wave / wps;
    const int sub = wave % wps;
    const bool active = node < p.N;
    const int nclamp = active ? node : 0;
    int seg_s = p.seg_ptr[nclamp];
    int seg_t = p.seg_ptr[nclamp + 1];
    const int half = lane >> 5, ch = lane & 31;
    float psrc[kEF];
    float cinit = 0.f;
    float bw[3] = {0.f, 0.f, 0.f};
    MsgB mb;   // 'sum' / 'mean': the message on the bf16 matrix pipe in split form (msg_bf16.cuh), the arithmetic of the fast kernel
    {
        const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = p.update ? psq[f] : 0.f;
        if (MSG) {
            cinit = psq[8 + ch];
#pragma unroll
            for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
            if (!AGG_MAX && !p.msg_f32) {
                msg_b_weights(blob + p.off_wnebf, lane, mb);
                msg_b_bias(cinit, lane, mb);
            }
        }
    }
    if (MSG) {
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        for (int i = tid; i < p.hin * kProjOut / 4; i += 256) l4[i] = g4[i];
    }
    if (p.pd_lds) {  // the whole gather table rides along with the first round trip instead of costing a dependent one
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.pd_in);
        f32x4* l4 = reinterpret_cast<f32x4*>(s_pd);
        for (int i = tid; i < p.N * (kPdStride / 4); i += 256) l4[i] = g4[i];
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {  // poisoned graph: make the failure visible in the outputs
        if (p.logits)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    if (MSG || p.pd_lds) __syncthreads();
    if (!active) seg_s = seg_t = 0;

    constexpr bool agg_max = AGG_MAX;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = agg_max ? -INFINITY : 0.f;

    const float* __restrict__ wee = blob + p.off_wee;
    for (int base = seg_s + 64 * sub; base < seg_t; base += 64 * wps) {
        const int k = base + lane;
        const bool valid = k < seg_t;
        const int kk = valid ? k : seg_t - 1;
        const int ko = unsorted ? p.perm[kk] : kk;  // the caller's edge id
        float ein[EFIN];
        if (p.first) {
            // edge encoder: Linear(edge_in, 6) + ReLU on data.edge_attr (models/mpn.py:137)
            float a[kMaxEdgeIn];
            const float* __restrict__ ap = p.edge_attr + (size_t)ko * p.edge_in;
            if (p.attr_vec) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap);
                a[0] = v[0], a[1] = v[1], a[2] = v[2], a[3] = v[3];
            } else {
                for (int j = 0; j < p.edge_in; ++j) a[j] = ap[j];
            }
            float e0v[kEF];
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                float s = blob[p.off_encb + f];
                if (p.attr_vec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = fmaf(blob[p.off_encw + f * 4 + j], a[j], s);
                } else {
                    for (int j = 0; j < p.edge_in; ++j) s = fmaf(blob[p.off_encw + f * p.edge_in + j], a[j], s);
                }
                e0v[f] = fmaxf(s, 0.f);
                if (p.drop.p_enc > 0.f) e0v[f] *= drop_scale(*p.drop.seed, kDropEncEdge, (unsigned long long)ko * kEF + f, p.drop.p_enc);
            }
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                ein[f] = e0v[f];
                if (REATT_E) {
                    ein[kEF + f] = e0v[f];  // cat((initial, latent)) with latent == initial on step 1 (mpn.py:283)
                    if (valid) p.e0[(size_t)f * p.e_stride + k] = e0v[f];
                }
                if (p.trace_e_enc && valid) p.trace_e_enc[(size_t)ko * kEF + f] = e0v[f];
            }
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                if (REATT_E) {
                    ein[f] = p.e0[(size_t)f * p.e_stride + kk];
                    ein[kEF + f] = p.e[(size_t)f * p.e_stride + kk];
                } else {
                    ein[f] = p.e[(size_t)f * p.e_stride + kk];
                }
            }
        }
        float en[kEF];
        if (p.update) {
            // edge update: ReLU(W_e . cat(x[row], x[col], e) + b_e)   (models/mpn.py:48, 68-69)
            const int j = p.col32[kk];
            f32x4 pd0;
            f32x2 pd1;
            if (p.pd_lds) {
                pd0 = *reinterpret_cast<const f32x4*>(s_pd + j * kPdStride);
                pd1 = *reinterpret_cast<const f32x2*>(s_pd + j * kPdStride + 4);
            } else {
                const float* __restrict__ pdj = p.pd_in + (size_t)j * kPdStride;
                pd0 = *reinterpret_cast<const f32x4*>(pdj);
                pd1 = *reinterpret_cast<const f32x2*>(pdj + 4);
            }
            const float pd[kEF] = {pd0[0], pd0[1], pd0[2], pd0[3], pd1[0], pd1[1]};
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                float s = psrc[f] + pd[f];
#pragma unroll
                for (int g = 0; g < EFIN; ++g) s = fmaf(wee[f * EFIN + g], ein[g], s);
                en[f] = fmaxf(s, 0.f);
                if (p.drop.p_edge > 0.f)
                    en[f] *= drop_scale(*p.drop.seed, kDropEdgeStep + p.step_no, (unsigned long long)ko * kEF + f, p.drop.p_edge);
            }
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) en[f] = ein[EFIN - kEF + f];
        }
        if (valid) {
            if (p.store_e) {
#pragma unroll
                for (int f = 0; f < kEF; ++f) p.e[(size_t)f * p.e_stride + k] = en[f];
            }
            if (p.trace_e) {
#pragma unroll
                for (int f = 0; f < kEF; ++f) p.trace_e[(size_t)ko * kEF + f] = en[f];
            }
        }
        if (p.cls_layers != 0) {
            // classifier.edge_mlp (models/mpn.py:292): Linear(6,C1) [BN folded] ReLU Linear(C1,1), or Linear(6,1)
            float logit;
            if (p.cls_layers == 2) {
                logit = blob[p.off_cb2];
                for (int q = 0; q < p.cls_hidden; ++q) {
                    float z = blob[p.off_cb1 + q];
#pragma unroll
                    for (int f = 0; f < kEF; ++f) z = fmaf(blob[p.off_cw1 + q * kEF + f], en[f], z);
                    z = fmaxf(z, 0.f);
                    if (p.drop.p_cls > 0.f)
                        z *= drop_scale(*p.drop.seed, kDropCls + p.cls_no, (unsigned long long)ko * p.cls_hidden + q, p.drop.p_cls);
                    logit = fmaf(blob[p.off_cw2 + q], z, logit);
                }
            } else {
                logit = blob[p.off_cb1];
#pragma unroll
                for (int f = 0; f < kEF; ++f) logit = fmaf(blob[p.off_cw1 + f], en[f], logit);
            }
            if (valid) p.logits[ko] = logit;
        }
        if (MSG) {
            // node message: ReLU(W_n . cat(x[row], e') + b_n)   (models/mpn.py:97-98), 64 edges x 32 channels
            f32x16 d0, d1;
            if (!AGG_MAX && !p.msg_f32) {
                MsgA oa;
                msg_a_operands(en, base, seg_t, lane, oa);
                d0 = msg_tile(oa, mb, 0);
                d1 = msg_tile(oa, mb, 1);
            } else {   // small graphs (msg_f32) and 'max': an ordered fp32 FMA chain, which the backward's arg-max pass recomputes bit for bit (backward.cuh)
#pragma unroll
                for (int i = 0; i < 16; ++i) d0[i] = d1[i] = cinit;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    // lanes = edges.  After the swap: r[0] = A operand of tile 0 (edges 0..31), r[1] = of tile 1.
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(en[2 * s]), __float_as_uint(en[2 * s + 1]),
                                                                    false, false);
                    d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[0]), bw[s], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[1]), bw[s], d1, 0, 0, 0);
                }
            }
            // accumulator register i of lane (ch, half) is edge (i&3) + 8*(i>>2) + 4*half of the tile
            if (p.drop.p_node > 0.f) {
                // train-mode Dropout on the messages (node_mlp's output, before aggregation): element (edge ko, channel ch)
                const unsigned long long seed = *p.drop.seed;
                const int rem = seg_t - base;
                const float ident = agg_max ? -INFINITY : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int eo = (i & 3) + 8 * (i >> 2) + 4 * half;
                    const int ko0 = __shfl(ko, eo), ko1 = __shfl(ko, eo + 32);   // caller's ids of the two tiles' edges
                    float m0 = fmaxf(d0[i], 0.f) * drop_scale(seed, kDropNodeStep + p.step_no, (unsigned long long)ko0 * kH + ch, p.drop.p_node);
                    float m1 = fmaxf(d1[i], 0.f) * drop_scale(seed, kDropNodeStep + p.step_no, (unsigned long long)ko1 * kH + ch, p.drop.p_node);
                    if (eo >= rem) m0 = ident;
                    if (eo + 32 >= rem) m1 = ident;
                    acc[i] = agg_max ? fmaxf(acc[i], fmaxf(m0, m1)) : (acc[i] + m0) + m1;
                }
            } else if (base + 64 <= seg_t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float m0 = fmaxf(d0[i], 0.f), m1 = fmaxf(d1[i], 0.f);
                    acc[i] = agg_max ? fmaxf(acc[i], fmaxf(m0, m1)) : (acc[i] + m0) + m1;
                }
            } else {
                const int rem = seg_t - base - 4 * half;
                const float ident = agg_max ? -INFINITY : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int eo = (i & 3) + 8 * (i >> 2);
                    const float m0 = (eo < rem) ? fmaxf(d0[i], 0.f) : ident;
                    const float m1 = (eo + 32 < rem) ? fmaxf(d1[i], 0.f) : ident;
                    acc[i] = agg_max ? fmaxf(acc[i], fmaxf(m0, m1)) : (acc[i] + m0) + m1;
                }
            }
        }
    }

    if (MSG) {
        // aggregate by SOURCE node (models/mpn.py:99, 192-202): registers -> half-waves -> waves of the group
        float v;
        if (agg_max) {
            v = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) v = fmaxf(v, acc[i]);
            v = fmaxf(v, __shfl_xor(v, 32));
        } else {
            v = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += acc[i];
            v += __shfl_xor(v, 32);
        }
        if (wps > 1) {
            if (lane < kH) s_part[wave * kH + lane] = v;
            __syncthreads();
            if (sub == 0) {
                const int w0 = wave;
                v = s_part[w0 * kH + ch];
                for (int u = 1; u < wps; ++u) {
                    const float o = s_part[(w0 + u) * kH + ch];
                    v = agg_max ? fmaxf(v, o) : v + o;
                }
            }
        }
        if (active && sub == 0) {
            const int deg = seg_t - seg_s;
            if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);  // scatter_mean: count clamped to 1
            if (deg == 0) v = 0.f;                                      // rows that receive nothing are 0
            if (p.trace_h && lane < kH) p.trace_h[(size_t)node * kH + lane] = v;
            if (p.pd_out) {
                const float hi = p.reatt_n ? p.h0[(size_t)node * kH + ch] : 0.f;
                project_node(v, hi, p.reatt_n != 0, s_proj, blob + p.off_projb, p.pd_out + (size_t)node * kPdStride,
                             p.psq_out + (size_t)node * kPsQStride, lane);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------

template <bool RE, bool MSG, bool MX>
static hipError_t launch_step_t(const StepParams& sp, hipStream_t st) {
    const int npg = 4 / sp.wps;
    const unsigned blocks = (unsigned)((sp.N + npg - 1) / npg);
    const size_t lds = ((MSG ? (size_t)sp.hin * kProjOut : 0) + 4 * kH + (sp.pd_lds ? (size_t)sp.N * kPdStride : 0)) * sizeof(float);
    GNNCCA_LAUNCH((mpn_step_kernel<RE, MSG, MX>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool RE, bool MSG>
static hipError_t launch_step(const StepParams& sp, hipStream_t st) {
    return sp.agg == GNNCCA_AGG_MAX ? launch_step_t<RE, MSG, true>(sp, st) : launch_step_t<RE, MSG, false>(sp, st);
}



}  // namespace gnncca
