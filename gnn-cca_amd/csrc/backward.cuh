#pragma once
// backward pass of the message-passing path (training through the HIP kernels).
//
// Forward quantities are NOT re-derived from scratch: the traced forward saves the latents the reference's autograd
// would keep alive (encoder outputs, node / edge latents after every step); ReLU masks come from those (y > 0), the
// node-message pre-activation is recomputed per edge from a per-node table Q = h W_nx^T + b_n.  Per step, in reverse:
//   classifier      (models/mpn.py:292)   d logits -> d e_s, d W_cls
//   node update     (models/mpn.py:97-99) d h_s[row] (/ deg for 'mean') -> ReLU' -> d W_ne, d b_n, d Q[row], d e_s
//   edge update     (models/mpn.py:48,68-69) d e_s -> ReLU' -> d W_ee, d b_e, d P_src[row], d P_dst[col], d e_{s-1}
//   projections     d (P_src | P_dst | Q) -> d h_{s-1}, d W_src, d W_dst, d W_nx
// then the two encoders.  Parameter gradients are sums over edges / nodes: per-edge contributions are reduced across
// the wave sixteen at a time (transposing butterfly, wave_lds_add16) into workgroup-resident LDS sums that reach
// global memory with one float atomic per value and (persistent) workgroup; sums of outer products over nodes run on
// the fp32 MFMA pipe (bwd_outer_mfma_kernel).  The result depends
// on arrival order in the last bits exactly like the reference on CUDA (cuBLAS / torch_scatter atomics).
// Supported: the MFMA family -- both reattach flags, all three aggregators -- with BatchNorm nowhere or inside the
// classifier only and a two-layer node encoder.
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one atomic per wave for a contribution every lane holds
__device__ __forceinline__ void wave_atomic_add(float* dst, float v) {
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(dst, v);
}

__global__ __launch_bounds__(256) void bwd_degree_kernel(const long long* __restrict__ ei, long long E, int N, int* __restrict__ deg) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    const long long r = ei[k];
    if (r >= 0 && r < N) atomicAdd(&deg[r], 1);
}

// Q[i][c] = b_n[c] + sum_d W_n[c][d] hin[i][d]   (W_n row-major [32][HI + 6]; hin = h, or cat(h0, h) with
// reattach_initial_nodes: HI = 64, models/mpn.py:285)
__global__ __launch_bounds__(256) void bwd_q_kernel(const float* __restrict__ h0, const float* __restrict__ h,
                                                    const float* __restrict__ Wn, const float* __restrict__ bn,
                                                    float* __restrict__ Q, int N, int HI) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * kH) return;
    const int i = t / kH, c = t - i * kH;
    const float* __restrict__ w = Wn + (size_t)c * (HI + kEF);
    float acc = bn[c];
    if (HI > kH) {
#pragma unroll
        for (int d = 0; d < kH; ++d) acc = fmaf(w[d], h0[(size_t)i * kH + d], acc);
        w += kH;
    }
#pragma unroll
    for (int d = 0; d < kH; ++d) acc = fmaf(w[d], h[(size_t)i * kH + d], acc);
    Q[t] = acc;
}

constexpr int kBwdLdsRows = 64;  // source rows a workgroup of bwd_edge_kernel accumulates in LDS

struct BwdEdgeParams {
    const long long* ei;
    const float* e_cur;    // [E][6] latent after this step
    const float* e_prev;   // [E][6] latent before this step (encoder output for step 1)
    const float* e0;       // [E][6] encoder output (reattach_initial_edges: the edge MLP reads cat(e0, e_prev)), or null
    float* ge0_acc;        // [E][6] d loss / d e0 through the reattached copies, accumulated over the steps, or null
    int HI;                // width of the node input of both MLPs: 32, or 64 with reattach_initial_nodes
    const float* Q;        // [N][32]
    const float* g_h;      // [N][32] d loss / d h_s, or null on the last step (its node update is dead)
    const int* deg;        // [N] (mean) or null
    const int* hmax;       // [N][32] ('max'): bit pattern of the largest positive message per node and channel, or null
    const int* hcnt;       // [N][32] ('max'): E - id of the FIRST edge that attains it (it alone receives the gradient), 0: none
    const float* g_logit;  // [E] or null
    const float* ge_in;    // [E][6] from step s+1, or null
    float* ge_out;         // [E][6] d loss / d e_{s-1}
    float* dP;             // [N][44]: [0,6) dP_src, [6,12) dP_dst, [12,44) dQ   (zeroed before the launch)
    const float* We;       // [6][70]
    const float* Wn;       // [32][38]
    const float* Wc1;      // [C1][6] or [1][6]
    const float* bc1;
    const float* Wc2;      // [1][C1] (two-layer classifier) or null
    float* gWe;            // [6][70]
    float* gbe;
    float* gWn;            // [32][38]
    float* gbn;
    float* gWc1;
    float* gbc1;
    float* gWc2;
    float* gbc2;
    // train-mode BatchNorm between the classifier's two layers (null when it has none): per hidden unit q
    const float* bn_gamma;
    const float* bn_stat;  // [C1][2] batch mean, 1/sqrt(batch var + eps) of this step's pre-activation
    const float* bn_red;   // [C1][2] mean over edges of g_y and of g_y * z_hat (from bwd_cls_bn_reduce_kernel)
    const float* bn_beta;
    long long E;
    int N, cls_hidden;     // cls_hidden == 0: single Linear(6,1)
    DropCfg drop;          // train-mode Dropout of this iteration (p == 0: off); masks are re-derived, never loaded
    int step_no, cls_no;   // 1-based step / 0-based classified-step index: the dropout streams of this launch
};

// 'max' aggregation, pass 1: hmax[i][c] = bit pattern of max over the edges of node i of the (positive) message
// pre-activation b = Q[i][c] + W_ne[c] . e', recomputed with exactly the arithmetic bwd_edge_kernel uses, so that its
// equality test selects the same edge.  Positive floats order like their bit patterns: atomicMax on int.
// Pass 2 (COUNT): hcnt[i][c] = E - (id of the first edge that attains it): torch_scatter's scatter_max hands the whole gradient to
// its `arg`, the first source row with the maximum (its CPU reducer updates on a strict `>`).
template <bool COUNT>
__global__ __launch_bounds__(256) void bwd_max_kernel(const long long* __restrict__ ei, const float* __restrict__ e_cur,
                                                      const float* __restrict__ Q, const float* __restrict__ Wn_, long long E,
                                                      int N, int HI, int* __restrict__ hmax, int* __restrict__ hcnt,
                                                      DropCfg drop, int step_no) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* Wn = (cfloat*)(unsigned long long)Wn_;
    const long long ri = ei[k];
    if ((unsigned long long)ri >= (unsigned long long)N) return;  // bad index: the forward flagged it and poisoned the logits
    const int i = (int)ri;
    float es[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) es[f] = e_cur[k * kEF + f];
    for (int c = 0; c < kH; ++c) {
        float b = Q[(size_t)i * kH + c];
#pragma unroll
        for (int f = 0; f < kEF; ++f) b = fmaf(Wn[c * (HI + kEF) + HI + f], es[f], b);
        if (drop.p_node > 0.f)   // the maximum is taken over the messages AFTER Dropout (0 for a dropped one)
            b = fmaxf(b, 0.f) * drop_scale(*drop.seed, kDropNodeStep + step_no, (unsigned long long)k * kH + c, drop.p_node);
        if (b > 0.f) {
            if (COUNT) {  // pass 2: the FIRST edge (lowest id k) that attains the maximum, kept as E - k so that 0 means "none"
                if (__float_as_int(b) == hmax[(size_t)i * kH + c]) atomicMax(&hcnt[(size_t)i * kH + c], (int)(E - k));
            } else {
                atomicMax(&hmax[(size_t)i * kH + c], __float_as_int(b));
            }
        }
    }
}

// Workgroup-resident accumulators of the parameter gradients this kernel produces (LDS slots):
constexpr int kSlotWe = 0;                    // [6][12] d W_ee  (12 columns with reattach_initial_edges, else 6 used)
constexpr int kSlotBe = kSlotWe + 72;         // [6]     d b_e
constexpr int kSlotWn = kSlotBe + 6;          // [32][6] d W_ne
constexpr int kSlotBn = kSlotWn + 192;        // [32]    d b_n
constexpr int kSlotWc1 = kSlotBn + 32;        // [C1][6] d W_cls1   (C1 <= kMaxCls; single Linear: [1][6])
constexpr int kSlotBc1 = kSlotWc1 + 6 * kMaxCls;
constexpr int kSlotWc2 = kSlotBc1 + kMaxCls;  // [C1]    d W_cls2
constexpr int kSlotBc2 = kSlotWc2 + kMaxCls;  // [1]
constexpr int kBwdSlots = kSlotBc2 + 1;

// Sixteen contributions every lane holds, each summed over the wave and added to its own LDS slot: one transposing
// reduction (17 cross-lane operations instead of 96) and ONE ds_add_f32, issued by the 16 lanes 4 idx that end up
// holding the sums.  `slot_of(idx)` maps idx -> slot (-1: unused); it is evaluated per lane.
template <typename SlotOf>
__device__ __forceinline__ void wave_lds_add16(const float (&v)[16], float* s_acc, SlotOf slot_of) {
    const float z = transpose_reduce16(v);
    const int lane = threadIdx.x & 63;
    const int slot = slot_of(lane >> 2);
    if ((lane & 3) == 0 && slot >= 0 && z != 0.f) atomicAdd(&s_acc[slot], z);
}

// One thread per edge, a PERSISTENT grid: a workgroup walks 256-edge chunks and keeps its parameter-gradient sums in
// LDS for its whole lifetime, so global memory sees one atomic per parameter and workgroup instead of one per wave
// (every wave hammering the same 1.3 KB of gradient memory was what bounded the first version of this kernel).
__global__ __launch_bounds__(256) void bwd_edge_kernel(const BwdEdgeParams p) {
    __shared__ float s_acc[kBwdSlots];
    // Gradients of the per-node tables indexed by the SOURCE row (d P_src, d Q) are first summed in LDS: with rows
    // sorted (the usual edge order) a chunk's 256 edges touch a handful of rows, so the ~15-255 same-address global
    // atomics per row become LDS atomics plus ONE global atomic per (row, column) and chunk.  Rows further than
    // kBwdLdsRows from the chunk's first row (unsorted or very sparse input) go straight to global memory.
    __shared__ float s_dp[kBwdLdsRows * 44];
    for (int t = threadIdx.x; t < kBwdSlots; t += 256) s_acc[t] = 0.f;
    // The small read-only tensors are read through the CONSTANT address space: wave-uniform reads become s_load
    // into SGPRs and are not ordered against the atomics below.
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* We = (cfloat*)(unsigned long long)p.We;
    cfloat* Wn = (cfloat*)(unsigned long long)p.Wn;
    cfloat* Wc1 = (cfloat*)(unsigned long long)p.Wc1;
    cfloat* bc1 = (cfloat*)(unsigned long long)p.bc1;
    cfloat* Wc2 = (cfloat*)(unsigned long long)p.Wc2;
    cfloat* bn_gamma = (cfloat*)(unsigned long long)p.bn_gamma;
    cfloat* bn_beta = (cfloat*)(unsigned long long)p.bn_beta;
    cfloat* bn_stat = (cfloat*)(unsigned long long)p.bn_stat;
    cfloat* bn_red = (cfloat*)(unsigned long long)p.bn_red;
    const int lane = threadIdx.x & 63;
    const int HI = p.HI, WnLd = HI + kEF;                 // node-MLP weight rows: [x[row] (HI) | e' (6)]
    const int EI = p.e0 ? 2 * kEF : kEF;                  // edge part of the edge-MLP input: e, or cat(e0, e)
    const int WeLd = 2 * HI + EI, WeE = 2 * HI;           // edge-MLP weight rows: [x[row] (HI) | x[col] (HI) | edge (EI)]

    for (long long chunk = blockIdx.x; chunk * 256 < p.E; chunk += gridDim.x) {
        const long long k0 = chunk * 256 + threadIdx.x;
        const bool has = k0 < p.E;
        const long long k = has ? k0 : p.E - 1;  // every lane takes part in the wave reductions; invalid lanes add zeros
        // An edge with an index outside [0, N) is DEAD here (no read or atomic at a wild address): the forward has flagged
        // it (GNNCCA_GRAPH_BAD_INDEX) and poisoned the logits with NaN, so the loss and every gradient are NaN anyway --
        // the reference raises an IndexError at this point (models/mpn.py:48).
        const long long ri = p.ei[k], rj = p.ei[p.E + k], rf = p.ei[chunk * 256];
        const bool inb = (unsigned long long)ri < (unsigned long long)p.N && (unsigned long long)rj < (unsigned long long)p.N;
        const bool valid = has && inb;
        const float live = valid ? 1.f : 0.f;
        const int i = inb ? (int)ri : 0, j = inb ? (int)rj : 0;
        const int i_first = (unsigned long long)rf < (unsigned long long)p.N ? (int)rf : 0;
        for (int t = threadIdx.x; t < kBwdLdsRows * 44; t += 256) s_dp[t] = 0.f;
        __syncthreads();
        const unsigned li = (unsigned)(i - i_first);
        auto add_row = [&](int col, float v) {
            if (li < (unsigned)kBwdLdsRows)
                atomicAdd(&s_dp[li * 44 + col], v);
            else
                atomicAdd(&p.dP[(size_t)i * 44 + col], v);
        };
        // the row's Q and d h rows, requested up front (16 x 16 B per lane) so that the channel loop has no loads
        float q_row[kH], gh_row[kH];
        int hm_row[kH];
        if (p.g_h && p.hmax) {
#pragma unroll
            for (int c = 0; c < kH; ++c) hm_row[c] = p.hmax[(size_t)i * kH + c];
        }
        if (p.g_h) {
            const f32x4* __restrict__ q4 = reinterpret_cast<const f32x4*>(p.Q + (size_t)i * kH);
            const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.g_h + (size_t)i * kH);
#pragma unroll
            for (int c = 0; c < kH / 4; ++c) {
                const f32x4 a = q4[c], b = g4[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) q_row[4 * c + u] = a[u], gh_row[4 * c + u] = b[u];
            }
        }
        float es[kEF], ep[kEF], ez[kEF], ge[kEF];
#pragma unroll
        for (int f = 0; f < kEF; ++f) {
            es[f] = p.e_cur[k * kEF + f];
            ep[f] = p.e_prev[k * kEF + f];
            ez[f] = p.e0 ? p.e0[k * kEF + f] : 0.f;
            ge[f] = p.ge_in ? p.ge_in[k * kEF + f] * live : 0.f;
        }
        // ---- classifier ----------------------------------------------------------------------------------------
        if (p.g_logit) {
            const float dz = p.g_logit[k] * live;
            if (p.cls_hidden > 0 && p.bn_stat) {
                // Linear -> BatchNorm(batch statistics) -> ReLU -> Linear.  d W2, d b2, d gamma, d beta were formed by
                // bwd_cls_bn_reduce_kernel; here: g_z = gamma * invstd * (g_y - mean(g_y) - z_hat * mean(g_y z_hat))
                // two hidden units per reduction: slots [8u + 0] d b1[q0+u], [8u + 1 + f] d W1[q0+u][f]
                for (int q0 = 0; q0 < p.cls_hidden; q0 += 2) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = min(q0 + u, p.cls_hidden - 1);
                        float z1 = bc1[q];
#pragma unroll
                        for (int f = 0; f < kEF; ++f) z1 = fmaf(Wc1[q * kEF + f], es[f], z1);
                        const float zh = (z1 - bn_stat[2 * q]) * bn_stat[2 * q + 1];
                        const float y = fmaf(bn_gamma[q], zh, bn_beta[q]);
                        const float sc = p.drop.p_cls > 0.f ? drop_scale(*p.drop.seed, kDropCls + p.cls_no, (unsigned long long)k * p.cls_hidden + q,
                                                                         p.drop.p_cls) : 1.f;
                        const float gy = y > 0.f ? Wc2[q] * dz * sc : 0.f;
                        float gz1 = live * bn_gamma[q] * bn_stat[2 * q + 1] * (gy - bn_red[2 * q] - zh * bn_red[2 * q + 1]);
                        if (q0 + u >= p.cls_hidden) gz1 = 0.f;
                        v[8 * u] = gz1;
                        v[8 * u + 7] = 0.f;
#pragma unroll
                        for (int f = 0; f < kEF; ++f) {
                            v[8 * u + 1 + f] = gz1 * es[f];
                            ge[f] = fmaf(Wc1[q * kEF + f], gz1, ge[f]);
                        }
                    }
                    wave_lds_add16(v, s_acc, [&](int idx) {
                        const int q = q0 + (idx >> 3), r = idx & 7;
                        if (q >= p.cls_hidden || r == 7) return -1;
                        return r == 0 ? kSlotBc1 + q : kSlotWc1 + q * kEF + r - 1;
                    });
                }
            } else if (p.cls_hidden > 0) {
                {
                    const float s = wave_reduce_sum(dz);
                    if (lane == 0 && s != 0.f) atomicAdd(&s_acc[kSlotBc2], s);
                }
                // two hidden units per reduction: slots [8u + 0] d W2[q], [8u + 1] d b1[q], [8u + 2 + f] d W1[q][f]
                for (int q0 = 0; q0 < p.cls_hidden; q0 += 2) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = min(q0 + u, p.cls_hidden - 1);
                        const bool on = q0 + u < p.cls_hidden;
                        float z1 = bc1[q];
#pragma unroll
                        for (int f = 0; f < kEF; ++f) z1 = fmaf(Wc1[q * kEF + f], es[f], z1);
                        const float sc = p.drop.p_cls > 0.f ? drop_scale(*p.drop.seed, kDropCls + p.cls_no, (unsigned long long)k * p.cls_hidden + q,
                                                                         p.drop.p_cls) : 1.f;
                        const float gz1 = (on && z1 > 0.f) ? Wc2[q] * dz * sc : 0.f;
                        v[8 * u] = on ? dz * fmaxf(z1, 0.f) * sc : 0.f;
                        v[8 * u + 1] = gz1;
#pragma unroll
                        for (int f = 0; f < kEF; ++f) {
                            v[8 * u + 2 + f] = gz1 * es[f];
                            ge[f] = fmaf(Wc1[q * kEF + f], gz1, ge[f]);
                        }
                    }
                    wave_lds_add16(v, s_acc, [&](int idx) {
                        const int q = q0 + (idx >> 3), r = idx & 7;
                        if (q >= p.cls_hidden) return -1;
                        return r == 0 ? kSlotWc2 + q : (r == 1 ? kSlotBc1 + q : kSlotWc1 + q * kEF + r - 2);
                    });
                }
            } else {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = 0.f;
                v[0] = dz;
#pragma unroll
                for (int f = 0; f < kEF; ++f) {
                    v[1 + f] = dz * es[f];
                    ge[f] = fmaf(Wc1[f], dz, ge[f]);
                }
                wave_lds_add16(v, s_acc, [&](int idx) { return idx == 0 ? kSlotBc1 : (idx <= kEF ? kSlotWc1 + idx - 1 : -1); });
            }
        }
        // ---- node update -----------------------------------------------------------------------------------------
        if (p.g_h) {
            const float inv = p.deg ? 1.f / (float)max(p.deg[i], 1) : 1.f;
            // two channels per reduction: slots [8u + 0] d b_n[c0+u], [8u + 1 + f] d W_ne[c0+u][f]
#pragma unroll
            for (int c0 = 0; c0 < kH; c0 += 2) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = c0 + u;
                    float b = q_row[c];
#pragma unroll
                    for (int f = 0; f < kEF; ++f) b = fmaf(Wn[c * WnLd + HI + f], es[f], b);
                    // 'max': only the edge torch_scatter's `arg` names -- the first one that attains the node's maximum -- passes the
                    // gradient on (a maximum of 0, i.e. no positive message, passes nothing: ReLU' = 0 there anyway).  Ties are real
                    // (every edge whose six features are dead has b = Q[row][c]), which edge of a tie gets the gradient only
                    // matters for the per-edge terms: the parameter gradients of tied DEAD or DUPLICATE edges are the same either way.
                    float gb = 0.f;
                    const float scn = p.drop.p_node > 0.f ? drop_scale(*p.drop.seed, kDropNodeStep + p.step_no, (unsigned long long)k * kH + c,
                                                                       p.drop.p_node) : 1.f;
                    if (p.drop.p_node > 0.f) b = fmaxf(b, 0.f) * scn;   // the message as aggregated: after ReLU and Dropout
                    if (b > 0.f) {
                        if (p.hmax == nullptr)
                            gb = gh_row[c] * inv * live * scn;
                        else if (__float_as_int(b) == hm_row[c] && p.hcnt[(size_t)i * kH + c] == (int)(p.E - k))
                            gb = gh_row[c] * live * scn;
                    }
                    v[8 * u] = gb;
                    v[8 * u + 7] = 0.f;
#pragma unroll
                    for (int f = 0; f < kEF; ++f) {
                        v[8 * u + 1 + f] = gb * es[f];
                        ge[f] = fmaf(Wn[c * WnLd + HI + f], gb, ge[f]);
                    }
                    if (gb != 0.f) add_row(12 + c, gb);
                }
                wave_lds_add16(v, s_acc, [&](int idx) {
                    const int c = c0 + (idx >> 3), r = idx & 7;
                    if (r == 7) return -1;
                    return r == 0 ? kSlotBn + c : kSlotWn + c * kEF + r - 1;
                });
            }
        }
        // ---- edge update -----------------------------------------------------------------------------------------
        const float inv_keep_edge = 1.f / (1.f - p.drop.p_edge);
        float ga[kEF];
#pragma unroll
        for (int f = 0; f < kEF; ++f) {
            ga[f] = es[f] > 0.f ? ge[f] * inv_keep_edge : 0.f;   // es > 0 <=> positive AND kept (the saved latent is post-Dropout)
            if (ga[f] != 0.f) {
                add_row(f, ga[f]);
                atomicAdd(&p.dP[(size_t)j * 44 + 6 + f], ga[f]);
            }
        }
        if (p.e0 == nullptr) {
            // two output features per reduction: slots [8u + 0] d b_e[f0+u], [8u + 1 + g] d W_ee[f0+u][g]
#pragma unroll
            for (int f0 = 0; f0 < kEF; f0 += 2) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    v[8 * u] = ga[f0 + u];
                    v[8 * u + 7] = 0.f;
#pragma unroll
                    for (int g = 0; g < kEF; ++g) v[8 * u + 1 + g] = ga[f0 + u] * ep[g];
                }
                wave_lds_add16(v, s_acc, [&](int idx) {
                    const int f = f0 + (idx >> 3), r = idx & 7;
                    if (r == 7) return -1;
                    return r == 0 ? kSlotBe + f : kSlotWe + f * 12 + r - 1;
                });
            }
        } else {
            // reattach_initial_edges: the edge part of the input is cat(e0, e_prev), twelve columns -- one output feature
            // per reduction: slots [0] d b_e[f], [1 + g] d W_ee[f][g] (g < 6: e0, g >= 6: e_prev)
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                float v[16];
                v[0] = ga[f];
#pragma unroll
                for (int g = 0; g < kEF; ++g) v[1 + g] = ga[f] * ez[g], v[7 + g] = ga[f] * ep[g];
                v[13] = v[14] = v[15] = 0.f;
                wave_lds_add16(v, s_acc, [&](int idx) { return idx == 0 ? kSlotBe + f : (idx <= 12 ? kSlotWe + f * 12 + idx - 1 : -1); });
            }
        }
        if (has) {
            // gradient of the edge part of the input: W_ee^T g_a (zeros for a dead edge); with reattach the first six
            // columns belong to e0
            const int off_prev = p.e0 ? kEF : 0;
#pragma unroll
            for (int g = 0; g < kEF; ++g) {
                float s = 0.f, s0 = 0.f;
#pragma unroll
                for (int f = 0; f < kEF; ++f) {
                    s = fmaf(We[f * WeLd + WeE + off_prev + g], ga[f], s);
                    if (p.e0) s0 = fmaf(We[f * WeLd + WeE + g], ga[f], s0);
                }
                p.ge_out[k * kEF + g] = s;
                if (p.e0) p.ge0_acc[k * kEF + g] += s0;  // this thread owns edge k: no atomic needed
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < kBwdLdsRows * 44; t += 256) {
            const float v = s_dp[t];
            const int r = i_first + t / 44;
            if (v != 0.f && r < p.N) atomicAdd(&p.dP[(size_t)r * 44 + t % 44], v);
        }
        __syncthreads();  // s_dp is cleared again at the top of the next chunk
    }
    // ---- the workgroup's parameter-gradient sums -> global memory, one atomic each --------------------------------
    __syncthreads();
    for (int t = threadIdx.x; t < kBwdSlots; t += 256) {
        const float v = s_acc[t];
        if (v == 0.f) continue;
        float* dst;
        if (t < kSlotBe) {
            if (t % 12 >= EI) continue;
            dst = p.gWe + (t / 12) * WeLd + WeE + t % 12;
        }
        else if (t < kSlotWn) dst = p.gbe + (t - kSlotBe);
        else if (t < kSlotBn) dst = p.gWn + ((t - kSlotWn) / kEF) * WnLd + HI + (t - kSlotWn) % kEF;
        else if (t < kSlotWc1) dst = p.gbn + (t - kSlotBn);
        else if (t < kSlotBc1) dst = p.gWc1 + (t - kSlotWc1);
        else if (t < kSlotWc2) dst = p.gbc1 + (t - kSlotBc1);
        else if (t < kSlotBc2) dst = p.gWc2 + (t - kSlotWc2);
        else dst = p.gbc2;
        atomicAdd(dst, v);
    }
}

// ---- classifier with train-mode BatchNorm1d (models/mlp.py:15, batch statistics over all E edges) ------------------
__device__ __forceinline__ double wave_reduce_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sums[q] += sum_k z, sums[C1 + q] += sum_k z^2   with z = W1 e + b1   (double accumulation)
__global__ __launch_bounds__(256) void cls_bn_stats_kernel(const float* __restrict__ e, long long E, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, int C1, double* __restrict__ sums) {
    const long long k0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = k0 < E;
    const long long k = valid ? k0 : E - 1;
    float es[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) es[f] = e[k * kEF + f];
    for (int q = 0; q < C1; ++q) {
        float z = b1[q];
#pragma unroll
        for (int f = 0; f < kEF; ++f) z = fmaf(W1[q * kEF + f], es[f], z);
        const double zd = valid ? (double)z : 0.0;
        const double s1 = wave_reduce_sum_d(zd), s2 = wave_reduce_sum_d(zd * zd);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sums[q], s1);
            atomicAdd(&sums[C1 + q], s2);
        }
    }
}

// stat[q] = (mean, 1/sqrt(biased var + eps)); running buffers updated like torch.nn.BatchNorm1d (momentum 0.1, unbiased var)
__global__ void cls_bn_finalize_kernel(const double* __restrict__ sums, long long E, int C1, float* __restrict__ stat,
                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int q = threadIdx.x;
    if (q >= C1) return;
    const double mean = sums[q] / (double)E;
    double var = sums[C1 + q] / (double)E - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[2 * q] = (float)mean;
    stat[2 * q + 1] = (float)(1.0 / sqrt(var + 1e-5));
    if (running_mean) {
        const double unbiased = E > 1 ? var * (double)E / (double)(E - 1) : var;
        running_mean[q] = (float)(0.9 * (double)running_mean[q] + 0.1 * mean);
        running_var[q] = (float)(0.9 * (double)running_var[q] + 0.1 * unbiased);
    }
}

__global__ __launch_bounds__(256) void cls_bn_apply_kernel(const float* __restrict__ e, long long E, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stat,
                                                           const float* __restrict__ W2, const float* __restrict__ b2, int C1,
                                                           float* __restrict__ logits, DropCfg drop, int cls_no) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    float es[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) es[f] = e[k * kEF + f];
    float logit = b2[0];
    for (int q = 0; q < C1; ++q) {
        float z = b1[q];
#pragma unroll
        for (int f = 0; f < kEF; ++f) z = fmaf(W1[q * kEF + f], es[f], z);
        float y = fmaxf(fmaf(gamma[q], (z - stat[2 * q]) * stat[2 * q + 1], beta[q]), 0.f);
        if (drop.p_cls > 0.f) y *= drop_scale(*drop.seed, kDropCls + cls_no, (unsigned long long)k * C1 + q, drop.p_cls);
        logit = fmaf(W2[q], y, logit);
    }
    logits[k] = logit;
}

// backward reductions of the BN classifier: sums[q] += sum g_y, sums[C1+q] += sum g_y z_hat; d W2, d b2 directly
__global__ __launch_bounds__(256) void bwd_cls_bn_reduce_kernel(const float* __restrict__ e, const float* __restrict__ g_logit,
                                                                long long E, const float* __restrict__ W1, const float* __restrict__ b1,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ stat, const float* __restrict__ W2, int C1,
                                                                double* __restrict__ sums, float* __restrict__ gW2,
                                                                float* __restrict__ gb2, DropCfg drop, int cls_no) {
    const long long k0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = k0 < E;
    const long long k = valid ? k0 : E - 1;
    float es[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) es[f] = e[k * kEF + f];
    const float dz = valid ? g_logit[k] : 0.f;
    wave_atomic_add(gb2, dz);
    for (int q = 0; q < C1; ++q) {
        float z = b1[q];
#pragma unroll
        for (int f = 0; f < kEF; ++f) z = fmaf(W1[q * kEF + f], es[f], z);
        const float zh = (z - stat[2 * q]) * stat[2 * q + 1];
        const float y = fmaf(gamma[q], zh, beta[q]);
        const float sc = drop.p_cls > 0.f ? drop_scale(*drop.seed, kDropCls + cls_no, (unsigned long long)k * C1 + q, drop.p_cls) : 1.f;
        wave_atomic_add(gW2 + q, dz * fmaxf(y, 0.f) * sc);
        const double gy = y > 0.f ? (double)(W2[q] * dz * sc) : 0.0;
        const double s1 = wave_reduce_sum_d(gy), s2 = wave_reduce_sum_d(gy * (double)zh);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sums[q], s1);
            atomicAdd(&sums[C1 + q], s2);
        }
    }
}

// red[q] = (mean g_y, mean g_y z_hat);  d beta = sum g_y, d gamma = sum g_y z_hat (accumulated over the classified steps)
__global__ void bwd_cls_bn_finalize_kernel(const double* __restrict__ sums, long long E, int C1, float* __restrict__ red,
                                           float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    const int q = threadIdx.x;
    if (q >= C1) return;
    red[2 * q] = (float)(sums[q] / (double)E);
    red[2 * q + 1] = (float)(sums[C1 + q] / (double)E);
    g_beta[q] += (float)sums[q];
    g_gamma[q] += (float)sums[C1 + q];
}

// d hin[i][c] = sum_f W_src[f][c] dP_src[i][f] + W_dst[f][c] dP_dst[i][f] + sum_o W_nx[o][c] dQ[i][o],  c < HI.
// hin = h_{s-1} (HI = 32), or cat(h0, h_{s-1}) with reattach_initial_nodes (HI = 64): columns 0..31 then belong to the
// encoder output h0 (accumulated over the steps in g_h0_acc), columns 32..63 to h_{s-1}.
__global__ __launch_bounds__(256) void bwd_node_kernel(const float* __restrict__ dP, const float* __restrict__ We,
                                                       const float* __restrict__ Wn, float* __restrict__ g_h_prev,
                                                       float* __restrict__ g_h0_acc, int N, int HI, int WeLd) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * HI) return;
    const int i = t / HI, c = t - i * HI;
    const float* __restrict__ d = dP + (size_t)i * 44;
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < kEF; ++f) {
        acc = fmaf(We[f * WeLd + c], d[f], acc);
        acc = fmaf(We[f * WeLd + HI + c], d[6 + f], acc);
    }
#pragma unroll
    for (int o = 0; o < kH; ++o) acc = fmaf(Wn[o * (HI + kEF) + c], d[12 + o], acc);
    if (HI == kH)
        g_h_prev[t] = acc;
    else if (c < kH)
        g_h0_acc[(size_t)i * kH + c] += acc;  // one thread per element and launch: no atomic needed
    else
        g_h_prev[(size_t)i * kH + c - kH] = acc;
}

// a[t] += b[t]
__global__ __launch_bounds__(256) void bwd_add_kernel(float* __restrict__ a, const float* __restrict__ b, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) a[t] += b[t];
}

// Parameter gradients that are sums of outer products over rows (nodes or edges):
//   out[o][k] += sum_n A[n][o] * B[n][k]          and optionally      colsum[o] += sum_n A[n][o]
// on the fp32 MFMA pipe.  Both operands of v_mfma_f32_32x32x2_f32 are read straight from global memory in their
// natural layout -- lane (l % 32, l / 32) holds A[n + l/32][o0 + l%32] and B[n + l/32][k0 + l%32] -- so there is no
// LDS staging: the reduction index n (rows) is the MFMA's k, two rows per instruction.  One wave owns a 32 x 64
// output tile over a chunk of `rows_per_chunk` rows; partial tiles are added to the output with one atomic per element.
// The column sum comes from a third MFMA against a B that is 1 in column 0 (k-tile 0 only).
// Rows of the product may be routed to up to three destination matrices (the per-node gradient table dP holds
// d P_src | d P_dst | d Q side by side, and their products with h go to three different weight blocks).
struct OuterOut {
    float* ptr[3];
    int ld[3];
    int row_begin[4];  // rows [row_begin[j], row_begin[j+1]) of the product -> ptr[j] (nullptr: dropped)
};

__global__ __launch_bounds__(64) void bwd_outer_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                            int ldb, const OuterOut out, float* __restrict__ colsum, int N, int O,
                                                            int K, int rows_per_chunk) {
    const int lane = threadIdx.x, half = lane >> 5, l32 = lane & 31;
    const int k0 = blockIdx.x * 64, o0 = blockIdx.y * 32;
    const int n0 = blockIdx.z * rows_per_chunk, n1 = min(n0 + rows_per_chunk, N);
    const int o = o0 + l32;
    const bool o_ok = o < O, k_ok0 = k0 + l32 < K, k_ok1 = k0 + 32 + l32 < K;
    const bool want_cs = colsum != nullptr && blockIdx.x == 0;
    const float* __restrict__ pa = A + (o_ok ? o : 0);
    const float* __restrict__ pb0 = B + (k_ok0 ? k0 + l32 : 0);
    const float* __restrict__ pb1 = B + (k_ok1 ? k0 + 32 + l32 : 0);
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = 0.f;
    const float one = l32 == 0 ? 1.f : 0.f;
    for (int n = n0; n < n1; n += 16) {  // eight MFMA k-steps (16 rows) per round: 24 loads in flight
        float a[8], b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = n + 2 * u + half;
            const bool ok = r < n1;
            const size_t rr = (size_t)(ok ? r : n0);
            const float av = pa[rr * lda], bv0 = pb0[rr * ldb], bv1 = pb1[rr * ldb];
            a[u] = ok && o_ok ? av : 0.f;
            b0[u] = k_ok0 ? bv0 : 0.f;
            b1[u] = k_ok1 ? bv1 : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b1[u], acc1, 0, 0, 0);
            if (want_cs) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], one, acc2, 0, 0, 0);
        }
    }
    // D layout: lane (j = l % 32), register r -> row i = (r % 4) + 8 * (r / 4) + 4 * (l / 32)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oo = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (oo < O) {
            const int j = oo >= out.row_begin[2] ? 2 : (oo >= out.row_begin[1] ? 1 : 0);
            float* dst = out.ptr[j];
            if (dst != nullptr) {
                dst += (size_t)(oo - out.row_begin[j]) * out.ld[j] + k0 + l32;
                if (k_ok0 && acc0[r] != 0.f) atomicAdd(dst, acc0[r]);
                if (k_ok1 && acc1[r] != 0.f) atomicAdd(dst + 32, acc1[r]);
            }
            if (want_cs && l32 == 0 && acc2[r] != 0.f) atomicAdd(&colsum[oo], acc2[r]);
        }
    }
}

static hipError_t launch_outer_multi(const float* A, int lda, const float* B, int ldb, const OuterOut& out, float* colsum, int N,
                                     int O, int K, hipStream_t st) {
    // rows per wave: few output tiles -> short row chunks, so that ~1000 waves share the reduction over N
    const long long tiles = (long long)((K + 63) / 64) * ((O + 31) / 32);
    long long rows = ((long long)N * tiles / 1024 + 15) / 16 * 16;
    rows = std::max<long long>(16, std::min<long long>(rows, 256));
    // ... but at most ~96 row chunks: every chunk adds its tile into the SAME output elements with atomics, and those serialise per address
    // across the eight L2s (~15 ns each).  The layer-by-layer engine's edge-level gradients (19 200 rows into a 6 x 70 or 32 x 38 block) ran
    // 600 chunks: 46-114 us per launch, a third of its training step
    rows = std::max<long long>(rows, (((long long)N + 95) / 96 + 15) / 16 * 16);
    const dim3 grid((unsigned)((K + 63) / 64), (unsigned)((O + 31) / 32), (unsigned)((N + rows - 1) / rows));
    hipLaunchKernelGGL(bwd_outer_mfma_kernel, grid, dim3(64), 0, st, A, lda, B, ldb, out, colsum, N, O, K, (int)rows);
    return hipGetLastError();
}

static hipError_t launch_outer(const float* A, int lda, const float* B, int ldb, float* out, int ldo, float* colsum, int N, int O,
                               int K, hipStream_t st) {
    OuterOut oo;
    oo.ptr[0] = out, oo.ptr[1] = oo.ptr[2] = nullptr;
    oo.ld[0] = ldo, oo.ld[1] = oo.ld[2] = 0;
    oo.row_begin[0] = 0, oo.row_begin[1] = oo.row_begin[2] = oo.row_begin[3] = O;
    return launch_outer_multi(A, lda, B, ldb, oo, colsum, N, O, K, st);
}

// g[t] = y[t] > 0 ? g[t] : 0     (ReLU backward from the saved output)
// g *= (y > 0) * scale: ReLU' from the saved output; with Dropout the saved output is post-Dropout (y > 0 <=> positive and kept)
// and scale = 1 / (1 - p)
__global__ __launch_bounds__(256) void bwd_relu_mask_kernel(float* __restrict__ g, const float* __restrict__ y, long long n, float scale) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) g[t] = y[t] > 0.f ? g[t] * scale : 0.f;
}

// y[t] *= Dropout scale of element t of `stream` (re-derives the forward's mask on a recomputed activation)
__global__ __launch_bounds__(256) void apply_dropout_kernel(float* __restrict__ y, long long n, DropCfg drop, unsigned stream, float p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) y[t] *= drop_scale(*drop.seed, stream, (unsigned long long)t, p);
}

// out[n][k] = (y[n][k] > 0) * sum_o G[n][o] W[o][k]      (d activation of the previous layer)
__global__ __launch_bounds__(256) void bwd_matmul_mask_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                              const float* __restrict__ y, float* __restrict__ out, int N, int O,
                                                              int K, float scale) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)N * K) return;
    const int n = (int)(t / K), k = (int)(t - (long long)n * K);
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(G[(size_t)n * O + o], W[(size_t)o * K + k], acc);
    out[t] = y[t] > 0.f ? acc * scale : 0.f;
}

// edge encoder backward: g_e0 -> ReLU' -> d W_e0 [6][A], d b_e0.  Persistent grid with LDS-resident sums like
// bwd_edge_kernel (A <= 7; wider raw edge attributes take one atomic per wave and value).
__global__ __launch_bounds__(256) void bwd_edge_enc_kernel(const float* __restrict__ ge0, const float* __restrict__ e0,
                                                           const float* __restrict__ attr, int A, long long E,
                                                           float* __restrict__ gW, float* __restrict__ gb, float scale) {
    __shared__ float s_acc[kEF * 8];  // [f][0] d b[f], [f][1 + a] d W[f][a]
    for (int t = threadIdx.x; t < kEF * 8; t += 256) s_acc[t] = 0.f;
    __syncthreads();
    for (long long chunk = blockIdx.x; chunk * 256 < E; chunk += gridDim.x) {
        const long long k0 = chunk * 256 + threadIdx.x;
        const bool valid = k0 < E;
        const long long k = valid ? k0 : E - 1;
        if (A <= 7) {  // two output features per reduction: slots [8u + 0] d b[f0+u], [8u + 1 + a] d W[f0+u][a]
            float at[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) at[a] = a < A ? attr[k * A + a] : 0.f;
#pragma unroll
            for (int f0 = 0; f0 < kEF; f0 += 2) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = f0 + u;
                    const float g = (valid && e0[k * kEF + f] > 0.f) ? ge0[k * kEF + f] * scale : 0.f;
                    v[8 * u] = g;
#pragma unroll
                    for (int a = 0; a < 7; ++a) v[8 * u + 1 + a] = g * at[a];
                }
                wave_lds_add16(v, s_acc, [&](int idx) { return (idx & 7) > A ? -1 : (f0 + (idx >> 3)) * 8 + (idx & 7); });
            }
        } else {
            for (int f = 0; f < kEF; ++f) {
                const float g = (valid && e0[k * kEF + f] > 0.f) ? ge0[k * kEF + f] * scale : 0.f;
                wave_atomic_add(gb + f, g);
                for (int a = 0; a < A; ++a) wave_atomic_add(gW + f * A + a, g * attr[k * A + a]);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kEF * 8; t += 256) {
        const float v = s_acc[t];
        const int f = t >> 3, r = t & 7;
        if (v != 0.f) atomicAdd(r == 0 ? gb + f : gW + f * A + r - 1, v);
    }
}

}  // namespace gnncca
