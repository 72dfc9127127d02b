#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ============================================================================================================
// Generic family: any legal GRAPH_NET_PARAMS outside the MFMA family (node latent != 32, edge latent != 6,
// multi-layer edge / node MLPs, deeper classifiers ...).  None of the shipped configs needs it; it exists so
// that the module is a drop-in for the whole constructor contract (models/mpn.py:154-247).  It follows the
// reference op for op -- virtual concatenation, Linear(+folded BN)(+ReLU) layer by layer, aggregation over the
// CSR segments in the caller's edge order (the order torch's CPU index_add_ sums in).  Two kernels do all the work:
// an LDS-tiled dense layer (eight outputs per thread) and a wave-per-node ordered aggregator.
// ============================================================================================================
struct GenSeg {
    const float* ptr;   // [rows][ld]
    const int* idx;     // optional row gather (row32 / col32 in the caller's edge order), or null
    int ld, width;
};

// out[r][o] = [ReLU](b[o] + sum over the concatenated segments of W[o][:] . in[r][:])
// Weights arrive transposed and padded, Wt[k][OP] with OP = ceil8(O) (pack_generic).  A thread owns one row and EIGHT
// consecutive outputs; a workgroup = 256 / (OP / 8) rows.  The weights of a k-tile are staged in LDS once per
// workgroup (contiguous copy, k-major) and read back as two 16-B broadcasts per k; the row's inputs come straight
// from global memory (the OP / 8 threads of a row read the same address; consecutive k hit the same line).
constexpr int kGenTileFloats = 8192;  // 32 KB of LDS per k-tile
__global__ __launch_bounds__(256) void gen_dense_kernel(GenSeg s0, GenSeg s1, GenSeg s2, const float* __restrict__ Wt,
                                                        const float* __restrict__ b, float* __restrict__ out, long long M,
                                                        int K, int O, int ldw, int ld_out, int relu) {
    __shared__ __attribute__((aligned(16))) float s_w[kGenTileFloats];
    const int OP = (O + 7) / 8 * 8, OC = OP / 8;
    const int rows_per_block = 256 / OC;
    const int tid = threadIdx.x;
    const int rl = tid / OC, c = tid - rl * OC;
    const long long r = (long long)blockIdx.x * rows_per_block + rl;
    const bool live = rl < rows_per_block && r < M;
    const int kt_max = max(1, kGenTileFloats / OP);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = live ? b[8 * c + j] : 0.f;  // bias is padded to OP
    const GenSeg segs[3] = {s0, s1, s2};
    int koff = 0;
    for (int q = 0; q < 3; ++q) {
        const GenSeg sg = segs[q];
        if (sg.width == 0) continue;
        const long long rr = live ? (sg.idx ? (long long)sg.idx[r] : r) : 0;
        const float* __restrict__ src = sg.ptr + (size_t)rr * sg.ld;
        for (int k0 = 0; k0 < sg.width; k0 += kt_max) {
            const int kt = min(kt_max, sg.width - k0);
            __syncthreads();  // the previous tile is no longer being read
            // rows of Wt are ldw floats apart (ldw == OP unless this launch covers one column group of a wider layer)
            const int op4 = OP / 4;
            for (int i = tid; i < kt * op4; i += 256) {
                const int kk = i / op4, o4 = i - kk * op4;
                reinterpret_cast<f32x4*>(s_w)[i] = *reinterpret_cast<const f32x4*>(Wt + (size_t)(koff + k0 + kk) * ldw + 4 * o4);
            }
            __syncthreads();
            if (live) {
                for (int kk = 0; kk < kt; ++kk) {
                    const float xv = src[k0 + kk];
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_w + kk * OP + 8 * c);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_w + kk * OP + 8 * c + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = fmaf(w0[j], xv, acc[j]);
                        acc[4 + j] = fmaf(w1[j], xv, acc[4 + j]);
                    }
                }
            }
        }
        koff += sg.width;
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = 8 * c + j;
            if (o < O) out[(size_t)r * ld_out + o] = relu ? fmaxf(acc[j], 0.f) : acc[j];
        }
    }
}

__global__ __launch_bounds__(256) void gen_index32_kernel(const long long* __restrict__ ei, int E, int N,
                                                          int* __restrict__ row32, int* __restrict__ col32) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    const long long r = ei[k], c = ei[(size_t)E + k];
    row32[k] = (r >= 0 && r < N) ? (int)r : 0;  // out-of-range ids are flagged by the plan; keep gathers in bounds
    col32[k] = (c >= 0 && c < N) ? (int)c : 0;
}

__global__ __launch_bounds__(256) void gen_plan_finish_kernel(const long long* __restrict__ ei, int E, int N, int* seg_ptr,
                                                              int* col32, int* perm, int* cursor, unsigned* flags,
                                                              const unsigned* __restrict__ blockflags) {
    __shared__ unsigned smem[1024];
    plan_finish(ei, E, N, seg_ptr, col32, perm, cursor, flags, blockflags, smem);
}

// h[i][c] = agg over the segment of node i of m[k][c], k in the caller's edge order (models/mpn.py:99,192-202).
// One workgroup per node: its four waves take the four consecutive quarters of the segment, lane = channel (channels
// beyond 64 in further passes), every edge row one coalesced read, sixteen rows requested before they are added in
// order; the four partial results are then combined in order.  (A fixed association, reproducible run to run; it
// differs from torch's strictly sequential CPU sum in the last bit only.)
__global__ __launch_bounds__(256) void gen_aggregate_kernel(const float* __restrict__ m, const int* __restrict__ seg_ptr,
                                                            const int* __restrict__ perm, const unsigned* __restrict__ flags,
                                                            float* __restrict__ h, int N, int H, int agg) {
    __shared__ float s_part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x;
    const unsigned fl = flags[0];
    if (fl & GNNCCA_GRAPH_BAD_INDEX) {  // the plan is not trustworthy: poison, touch nothing else
        for (int c = threadIdx.x; c < H; c += 256) h[(size_t)i * H + c] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (fl & GNNCCA_GRAPH_UNSORTED) != 0;
    const int s = seg_ptr[i], e = seg_ptr[i + 1];
    const int len = e - s, q = (len + 3) / 4;
    const int ws = min(s + wave * q, e), we = min(ws + q, e);
    const float ident = agg == GNNCCA_AGG_MAX ? -INFINITY : 0.f;
    for (int c0 = 0; c0 < H; c0 += 64) {
        const int c = c0 + lane;
        float v = ident;
        if (c < H) {
            for (int p0 = ws; p0 < we; p0 += 16) {
                float x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int p = min(p0 + u, we - 1);
                    const int k = unsorted ? perm[p] : p;
                    x[u] = m[(size_t)k * H + c];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (p0 + u < we) v = agg == GNNCCA_AGG_MAX ? fmaxf(v, x[u]) : v + x[u];
            }
        }
        s_part[wave][lane] = v;
        __syncthreads();
        if (wave == 0 && c < H) {
            float r = s_part[0][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) r = agg == GNNCCA_AGG_MAX ? fmaxf(r, s_part[w][lane]) : r + s_part[w][lane];
            if (agg == GNNCCA_AGG_MEAN) r = r / (float)max(len, 1);
            if (len == 0) r = 0.f;
            h[(size_t)i * H + c] = r;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gen_poison_kernel(float* __restrict__ out, long long n, const unsigned* __restrict__ flags) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n && (flags[0] & GNNCCA_GRAPH_BAD_INDEX)) out[t] = __builtin_nanf("");
}

static int gen_run_mlp(const gnncca_mlp& mlp, const float* blob, const int32_t* woff, const int32_t* boff, GenSeg in0,
                       GenSeg in1, GenSeg in2, long long M, float* out_final, int ld_final, float* tmp_a, float* tmp_b,
                       int ld_tmp, hipStream_t st) {
    GenSeg none = {nullptr, nullptr, 0, 0};
    GenSeg a = in0, b = in1, c = in2;
    float* bufs[2] = {tmp_a, tmp_b};
    for (int l = 0; l < mlp.n_layers; ++l) {
        const gnncca_layer& L = mlp.layers[l];
        const bool last = l == mlp.n_layers - 1;
        float* dst = last ? out_final : bufs[l & 1];
        const int ld = last ? ld_final : ld_tmp;
        const long long total = M * L.out_dim;
        if (total > 0) {
            const int op_total = (L.out_dim + 7) / 8 * 8;
            for (int o0 = 0; o0 < L.out_dim; o0 += 2048) {  // column groups of at most 2048 outputs (256 threads x 8)
                const int og = std::min(2048, L.out_dim - o0);
                const int rows_per_block = 256 / ((og + 7) / 8);
                hipLaunchKernelGGL(gen_dense_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st,
                                   a, b, c, blob + woff[l] + o0, blob + boff[l] + o0, dst + o0, M, L.in_dim, og, op_total, ld,
                                   L.relu);
            }
            HIP_TRY(hipGetLastError());
        }
        a = GenSeg{dst, nullptr, ld, L.out_dim};
        b = c = none;
    }
    return GNNCCA_OK;
}

static int forward_generic(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                           const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                           float* logits_out, const gnncca_trace* trace, hipStream_t st) {
    const GenWorkspace ws = carve_generic(d, n_nodes, n_edges);
    if (workspace_bytes < ws.total) return GNNCCA_ERR_WORKSPACE;
    GenBlobHeader hdr;
    if (!gen_blob_header(d, &hdr)) return GNNCCA_ERR_UNSUPPORTED;
    const float* blob = static_cast<const float*>(packed_dev);
    char* base = static_cast<char*>(workspace);
    const int N = (int)n_nodes, E = (int)n_edges;
    const int H = d->node_dim, EF = d->edge_dim;
    unsigned* flags = reinterpret_cast<unsigned*>(base + ws.flags);
    unsigned* blockflags = reinterpret_cast<unsigned*>(base + ws.blockflags);
    int* seg_ptr = reinterpret_cast<int*>(base + ws.seg_ptr);
    int* col32 = reinterpret_cast<int*>(base + ws.col32);
    int* perm = reinterpret_cast<int*>(base + ws.perm);
    int* cursor = reinterpret_cast<int*>(base + ws.cursor);
    int* row32o = reinterpret_cast<int*>(base + ws.row32o);
    int* col32o = reinterpret_cast<int*>(base + ws.col32o);
    float* nb[3] = {reinterpret_cast<float*>(base + ws.node[0]), reinterpret_cast<float*>(base + ws.node[1]),
                    reinterpret_cast<float*>(base + ws.node[2])};
    // edge scratch: eb[0], eb[1] are layer temporaries; eb[2] / eb[3] alternate between "latent edge features" and
    // "per-edge messages" so that no launch reads and writes the same buffer
    float* eb[4] = {reinterpret_cast<float*>(base + ws.edge[0]), reinterpret_cast<float*>(base + ws.edge[1]),
                    reinterpret_cast<float*>(base + ws.edge[2]), reinterpret_cast<float*>(base + ws.edge[3])};
    float* h0 = reinterpret_cast<float*>(base + ws.h0);
    float* e0 = reinterpret_cast<float*>(base + ws.e0);
    const int nw = (int)ws.node_w, ew = (int)ws.edge_w;
    const GenSeg none = {nullptr, nullptr, 0, 0};
    const long long* ei = reinterpret_cast<const long long*>(edge_index);

    // the fused form of the message-passing steps: one launch per step (generic_fused.cuh) when every width fits its per-thread LDS budget
    int fT = 0, fW = 0;
    static const bool gen_unfused = diag_env("GNNCCA_GEN_UNFUSED") != nullptr;   // diagnostics: A/B against the op-by-op path
    const bool use_fused = !gen_unfused && gen_fused_ok(d, n_nodes, &fT, &fW);
    // graph plan (same kernels as the MFMA family: plan blocks, then one finishing workgroup)
    if (E > 0) {
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.ei = ei;
        ep.seg_ptr = seg_ptr;
        ep.col32 = col32;
        ep.blockflags = blockflags;
        ep.E = E;
        ep.N = N;
        hipLaunchKernelGGL(plan_only_kernel, dim3(plan_num_blocks(E)), dim3(256), 0, st, ep);
        HIP_TRY(hipGetLastError());
        if (!use_fused) {   // the op-by-op path gathers by the caller's edge order
            hipLaunchKernelGGL(gen_index32_kernel, dim3((E + 255) / 256), dim3(256), 0, st, ei, E, N, row32o, col32o);
            HIP_TRY(hipGetLastError());
        }
    }
    if (!use_fused) {   // (the fused form folds the plan's findings in its encoder-tail launch)
        hipLaunchKernelGGL(gen_plan_finish_kernel, dim3(1), dim3(256), 0, st, ei, E, N, seg_ptr, col32, perm, cursor, flags,
                           (const unsigned*)blockflags);
        HIP_TRY(hipGetLastError());
    }

    // encoder (models/mpn.py:270): node MLP on x, edge MLP on edge_attr
    int s;
    const int ks0 = gen_enc0_ksplit(d, n_nodes);
    GenStepParams gp;   // the fused form's per-step parameters (the part that does not change from step to step)
    std::memset(&gp, 0, sizeof(gp));
    float* tab[2] = {reinterpret_cast<float*>(base + ws.tab[0]), reinterpret_cast<float*>(base + ws.tab[1])};
    if (use_fused) {
        gp.blob = blob;
        gp.seg_ptr = seg_ptr, gp.col32 = col32, gp.perm = perm, gp.flags = flags;
        gp.N = N, gp.E = E;
        gp.H = H, gp.EF = EF, gp.agg = d->agg;
        gp.hin_w = d->reattach_nodes ? 2 * H : H;
        int wfl = 0;
        gen_fill_mlp(&gp.edge, d->edge_mlp, hdr.w[2], hdr.b[2], 2 * gp.hin_w, (d->reattach_edges ? 2 : 1) * EF, &wfl);   // the e block of cat[x[row] | x[col] | e]
        gen_fill_mlp(&gp.node, d->node_mlp, hdr.w[3], hdr.b[3], gp.hin_w, EF, &wfl);                                  // the e' block of cat[x[row] | e']
        gen_fill_mlp(&gp.cls, d->cls_edge, hdr.w[4], hdr.b[4], -1, 0, &wfl);
        gen_fill_mlp(&gp.enc, d->enc_edge, hdr.w[1], hdr.b[1], -1, 0, &wfl);   // the edge encoder rides in step 1
        gp.w_floats = (wfl + 3) / 4 * 4;
        gp.w_used = wfl;
        gp.step_w = hdr.step_w;
        if (hdr.step_w == 0 || hdr.step_w_floats != wfl) return GNNCCA_ERR_UNSUPPORTED;   // (the packer and gen_fill_mlp lay the stage image out alike)
        gp.h0 = d->reattach_nodes ? h0 : nullptr;
        gp.o1e = d->edge_mlp.layers[0].out_dim, gp.o1n = d->node_mlp.layers[0].out_dim;
        gp.tab_ld = 2 * gp.o1e + gp.o1n;
        gp.lds_stride = fT + 1;
        gp.wmax = fW;
        gp.tab_out = tab[0];
        // encoder.node_mlp: the wide first layer on the MFMA family's split-K fp32 GEMM when it pays, everything behind it -- slab sum,
        // further layers, h0, step 1's projection tables, the plan's flag fold -- in ONE launch (gen_node_tail_kernel)
        GenTailParams tq;
        std::memset(&tq, 0, sizeof(tq));
        int dummy = 0;
        gen_fill_mlp(&tq.enc, d->enc_node, hdr.w[0], hdr.b[0], -1, 0, &dummy);
        tq.x = x, tq.node_in = d->node_in;
        tq.h0 = h0, tq.trace_h = trace ? trace->h_enc : nullptr;
        tq.ei = ei, tq.seg_ptr = seg_ptr, tq.col32 = col32, tq.perm = perm, tq.cursor = cursor, tq.flags = flags, tq.blockflags = blockflags, tq.E = E;
        int wenc = std::max(std::max(gp.hin_w, H), 1024);   // (plan_finish wants 3 KB of scratch)
        for (int l = 0; l < d->enc_node.n_layers; ++l) wenc = std::max(wenc, (int)d->enc_node.layers[l].out_dim);
        if (d->enc_node.n_layers > 0 && ks0 > 0) {
            const gnncca_layer& l0 = d->enc_node.layers[0];
            float* part = reinterpret_cast<float*>(base + ws.partial);
            int kslice = (l0.in_dim + ks0 - 1) / ks0;
            kslice = (kslice + 63) / 64 * 64;
            EncPlanParams ep;
            std::memset(&ep, 0, sizeof(ep));
            ep.in = x, ep.W = blob + hdr.enc0_rowmajor, ep.part = part;
            ep.M = N, ep.K = l0.in_dim, ep.O = l0.out_dim, ep.kslice = kslice;
            ep.vec_ok = (l0.in_dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
            ep.nrt = (N + 31) / 32;
            ep.nks = ks0;
            ep.gemm_blocks = ep.nrt * ks0 * ((l0.out_dim + 127) / 128);
            hipLaunchKernelGGL(enc_gemm_plan_kernel, dim3(ep.gemm_blocks), dim3(256), 0, st, ep);
            HIP_TRY(hipGetLastError());
            tq.part = part, tq.ks = ks0, tq.first_layer = 1;
        } else {
            tq.part = nullptr, tq.ks = 0, tq.first_layer = 0;
            wenc = std::max(wenc, d->node_in);
        }
        tq.wmax_enc = wenc;
        const size_t tl = (size_t)2 * wenc * sizeof(float);
        if (tl > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_node_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl));
        hipLaunchKernelGGL(gen_node_tail_kernel, dim3((unsigned)N + 1), dim3(256), tl, st, gp, tq);
        HIP_TRY(hipGetLastError());
    } else if (d->enc_node.n_layers > 0 && ks0 > 0) {
        // first layer (the wide one: 2048-d embeddings) on the MFMA family's split-K fp32 GEMM + its reduce kernel,
        // the remaining layers on the tiled dense kernel
        const gnncca_layer& l0 = d->enc_node.layers[0];
        float* part = reinterpret_cast<float*>(base + ws.partial);
        int kslice = (l0.in_dim + ks0 - 1) / ks0;
        kslice = (kslice + 63) / 64 * 64;
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.in = x;
        ep.W = blob + hdr.enc0_rowmajor;
        ep.part = part;
        ep.M = N;
        ep.K = l0.in_dim;
        ep.O = l0.out_dim;
        ep.kslice = kslice;
        ep.vec_ok = (l0.in_dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        ep.nrt = (N + 31) / 32;
        ep.nks = ks0;
        ep.gemm_blocks = ep.nrt * ks0 * ((l0.out_dim + 127) / 128);
        hipLaunchKernelGGL(enc_gemm_plan_kernel, dim3(ep.gemm_blocks), dim3(256), 0, st, ep);
        HIP_TRY(hipGetLastError());
        const bool only = d->enc_node.n_layers == 1;
        float* dst0 = only ? h0 : nb[2];  // dense [N][out]
        hipLaunchKernelGGL(reduce_bias_act_kernel, grid1((size_t)N * l0.out_dim, 256), dim3(256), 0, st, (const float*)part,
                           blob + hdr.b[0][0], dst0, N, l0.out_dim, ks0, l0.relu);
        HIP_TRY(hipGetLastError());
        if (!only) {
            gnncca_mlp rest = d->enc_node;
            rest.n_layers = d->enc_node.n_layers - 1;
            for (int l = 0; l < rest.n_layers; ++l) rest.layers[l] = d->enc_node.layers[l + 1];
            s = gen_run_mlp(rest, blob, hdr.w[0] + 1, hdr.b[0] + 1, GenSeg{dst0, nullptr, l0.out_dim, l0.out_dim}, none, none, N,
                            h0, H, nb[0], nb[1], nw, st);
            if (s != GNNCCA_OK) return s;
        }
    } else if (d->enc_node.n_layers > 0) {
        s = gen_run_mlp(d->enc_node, blob, hdr.w[0], hdr.b[0], GenSeg{x, nullptr, d->node_in, d->node_in}, none, none, N, h0, H,
                        nb[0], nb[1], nw, st);
        if (s != GNNCCA_OK) return s;
    } else {
        HIP_TRY(hipMemcpyAsync(h0, x, (size_t)N * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (!use_fused && trace && trace->h_enc) HIP_TRY(hipMemcpyAsync(trace->h_enc, h0, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
    if (E == 0) return GNNCCA_OK;
    const bool enc_in_step1 = use_fused && d->enc_edge.n_layers > 0;   // the fused step 1 encodes the raw edge attributes itself
    if (enc_in_step1) {
        // nothing: e0 reaches HBM from step 1 only when a later step or the trace reads it
    } else if (d->enc_edge.n_layers > 0) {
        s = gen_run_mlp(d->enc_edge, blob, hdr.w[1], hdr.b[1], GenSeg{edge_attr, nullptr, d->edge_in, d->edge_in}, none, none, E,
                        e0, EF, eb[0], eb[1], ew, st);
        if (s != GNNCCA_OK) return s;
    } else {
        HIP_TRY(hipMemcpyAsync(e0, edge_attr, (size_t)E * EF * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (!enc_in_step1 && trace && trace->e_enc) HIP_TRY(hipMemcpyAsync(trace->e_enc, e0, (size_t)E * EF * 4, hipMemcpyDeviceToDevice, st));

    const int L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    const float* h_cur = h0;  // latent node feats (== initial before step 1); every h buffer is dense [N][H]
    const float* e_cur = e0;  // latent edge feats (== initial before step 1)
    int e_ld = EF;
    float* h_lat[2] = {nb[2], nb[0]};
    int out_idx = 0;
    auto classify_edges = [&](const float* ee, int ld) -> int {
        float* dst = logits_out + (size_t)(out_idx++) * E;
        int r = gen_run_mlp(d->cls_edge, blob, hdr.w[4], hdr.b[4], GenSeg{ee, nullptr, ld, EF}, none, none, E, dst, 1, eb[0],
                            eb[1], ew, st);
        if (r != GNNCCA_OK) return r;
        hipLaunchKernelGGL(gen_poison_kernel, grid1((size_t)E, 256), dim3(256), 0, st, dst, (long long)E, (const unsigned*)flags);
        return hipGetLastError() == hipSuccess ? GNNCCA_OK : GNNCCA_ERR_HIP;
    };
    if (L == 0) return classify_edges(e0, EF);
    // ---- the fused form: one launch per step (generic_fused.cuh) when every width fits its per-thread LDS budget -----------------------
    if (use_fused) {
        const size_t lds = gen_fused_lds_bytes(fW, fT, gp.w_floats, gp.tab_ld, gp.hin_w, H);
        if (lds > kGenFusedLdsMax) return GNNCCA_ERR_UNSUPPORTED;   // (gen_fused_ok admits by the same formula: unreachable)
        {   // (once per device and size: the call costs tens of microseconds of host time)
            static thread_local int attr_dev = -1;
            static thread_local size_t attr_lds = 0;
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            if (lds > 64 * 1024 && (attr_dev != dev || attr_lds < lds)) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_step_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_dev = dev, attr_lds = lds;
            }
        }
        for (int step = 1; step <= L; ++step) {
            float* e_new = eb[2 + (step & 1)];
            const bool need_h = step < L || (trace && trace->h_steps);
            float* hn = h_lat[step & 1];
            if (d->reattach_edges) {
                gp.e_a = e0, gp.e_a_ld = EF, gp.e_a_w = EF;
                gp.e_b = e_cur, gp.e_b_ld = e_ld, gp.e_b_w = EF;
            } else {
                gp.e_a = e_cur, gp.e_a_ld = e_ld, gp.e_a_w = EF;
                gp.e_b = nullptr, gp.e_b_ld = 0, gp.e_b_w = 0;
            }
            if (step == 1 && enc_in_step1) {
                gp.edge_attr = edge_attr, gp.edge_in = d->edge_in;
                gp.e0_out = (d->reattach_edges || (trace && trace->e_enc)) ? e0 : nullptr;
            } else {
                gp.edge_attr = nullptr, gp.e0_out = nullptr;
            }
            gp.tab_in = tab[(step - 1) & 1];
            gp.tab_out = step < L ? tab[step & 1] : nullptr;
            gp.e_new = e_new, gp.e_new_ld = ew;
            gp.logits = step >= first_cls ? logits_out + (size_t)(out_idx++) * E : nullptr;
            gp.h_new = need_h ? hn : nullptr;
            hipLaunchKernelGGL(gen_step_fused_kernel, dim3((unsigned)N), dim3((unsigned)fT), lds, st, gp);
            HIP_TRY(hipGetLastError());
            e_cur = e_new;
            e_ld = ew;
            if (step == 1 && enc_in_step1 && trace && trace->e_enc)
                HIP_TRY(hipMemcpyAsync(trace->e_enc, e0, (size_t)E * EF * 4, hipMemcpyDeviceToDevice, st));
            if (trace && trace->e_steps)
                HIP_TRY(hipMemcpy2DAsync(trace->e_steps + (size_t)(step - 1) * E * EF, (size_t)EF * 4, e_new, (size_t)ew * 4,
                                         (size_t)EF * 4, E, hipMemcpyDeviceToDevice, st));
            if (need_h && trace && trace->h_steps)
                HIP_TRY(hipMemcpyAsync(trace->h_steps + (size_t)(step - 1) * N * H, hn, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
        }
        return GNNCCA_OK;
    }
    for (int step = 1; step <= L; ++step) {
        float* e_new = eb[2 + (step & 1)];        // this step's latent edge features
        float* msg = eb[2 + ((step + 1) & 1)];    // this step's per-edge messages (the previous latent is dead by then)
        // x = cat(initial, latent) when reattach_initial_nodes (models/mpn.py:285): materialised in nb[1]
        const float* hin = h_cur;
        int hin_w = H, hin_ld = H;
        if (d->reattach_nodes) {
            float* cat = nb[1];
            HIP_TRY(hipMemcpy2DAsync(cat, (size_t)nw * 4, h0, (size_t)H * 4, (size_t)H * 4, N, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpy2DAsync(cat + H, (size_t)nw * 4, h_cur, (size_t)H * 4, (size_t)H * 4, N, hipMemcpyDeviceToDevice, st));
            hin = cat;
            hin_w = 2 * H;
            hin_ld = nw;
        }
        // e = cat(initial, latent) when reattach_initial_edges (models/mpn.py:283): materialised in eb[0]
        const float* ein = e_cur;
        int ein_w = EF, ein_ld = e_ld;
        float* tmp_a = eb[0];
        float* tmp_b = eb[1];
        if (d->reattach_edges) {
            HIP_TRY(hipMemcpy2DAsync(eb[0], (size_t)ew * 4, e0, (size_t)EF * 4, (size_t)EF * 4, E, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpy2DAsync(eb[0] + EF, (size_t)ew * 4, e_cur, (size_t)e_ld * 4, (size_t)EF * 4, E,
                                     hipMemcpyDeviceToDevice, st));
            ein = eb[0];
            ein_w = 2 * EF;
            ein_ld = ew;
            std::swap(tmp_a, tmp_b);  // layer 0 reads eb[0]: its output must go to eb[1]
        }
        // edge update: edge_mlp(cat[x[row], x[col], e])   (models/mpn.py:48, 68-69)
        s = gen_run_mlp(d->edge_mlp, blob, hdr.w[2], hdr.b[2], GenSeg{hin, row32o, hin_ld, hin_w},
                        GenSeg{hin, col32o, hin_ld, hin_w}, GenSeg{ein, nullptr, ein_ld, ein_w}, E, e_new, ew, tmp_a, tmp_b, ew,
                        st);
        if (s != GNNCCA_OK) return s;
        e_cur = e_new;
        e_ld = ew;
        if (trace && trace->e_steps)
            HIP_TRY(hipMemcpy2DAsync(trace->e_steps + (size_t)(step - 1) * E * EF, (size_t)EF * 4, e_new, (size_t)ew * 4,
                                     (size_t)EF * 4, E, hipMemcpyDeviceToDevice, st));
        // node update: aggregate over `row` of node_mlp(cat[x[row], e'])   (models/mpn.py:97-99)
        const bool need_h = step < L || (trace && trace->h_steps);
        if (need_h) {
            s = gen_run_mlp(d->node_mlp, blob, hdr.w[3], hdr.b[3], GenSeg{hin, row32o, hin_ld, hin_w},
                            GenSeg{e_new, nullptr, ew, EF}, none, E, msg, H, eb[0], eb[1], ew, st);
            if (s != GNNCCA_OK) return s;
            float* hn = h_lat[step & 1];
            hipLaunchKernelGGL(gen_aggregate_kernel, dim3((unsigned)N), dim3(256), 0, st, (const float*)msg,
                               (const int*)seg_ptr, (const int*)perm, (const unsigned*)flags, hn, N, H, d->agg);
            HIP_TRY(hipGetLastError());
            h_cur = hn;
            if (trace && trace->h_steps)
                HIP_TRY(hipMemcpyAsync(trace->h_steps + (size_t)(step - 1) * N * H, hn, (size_t)N * H * 4,
                                       hipMemcpyDeviceToDevice, st));
        }
        if (step >= first_cls) {
            s = classify_edges(e_new, ew);
            if (s != GNNCCA_OK) return s;
        }
    }
    return GNNCCA_OK;
}


}  // namespace gnncca
