#pragma once
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

// ============================================================================================================
// Layer-by-layer TRAINING engine (the part the fused train path of backward.cuh does not cover):
// every legal GRAPH_NET_PARAMS in train mode -- BatchNorm1d with batch statistics in ANY MLP (models/mlp.py:14-15), Dropout
// behind any ReLU (mlp.py:20-21), any widths and depths (the generic family), all three aggregators, both reattach flags.
// It follows train.py:454-494's forward + backward through models/mpn.py:250-299 op for op, as torch autograd would run them:
//   forward : Linear -> [BatchNorm (batch mean / biased variance, running buffers updated with momentum 0.1, unbiased variance)]
//             -> [ReLU -> Dropout], concatenations materialised, every activation autograd would keep goes to the TAPE;
//   backward: the same layers in reverse (ReLU / Dropout mask from the saved output, BatchNorm's two column reductions,
//             d W = d Z^T X on the MFMA outer-product kernel, d X = d Z W), scatter-adds for the gathered inputs, the
//             aggregator's backward (sum: broadcast; mean: / degree; max: torch_scatter's arg = FIRST edge attaining the maximum).
// Correctness first: one launch per op, fp32, double accumulators for the BatchNorm statistics (as torch's CPU kernels).
// The shipped shapes (BatchNorm nowhere or in the classifier only) keep the fused path; gnn-cca_amd/mpn.py picks.
// ============================================================================================================

constexpr float kBnEps = 1e-5f, kBnMomentum = 0.1f;

// dropout stream of layer `li` of an MLP call whose first layer has stream `base` (common.cuh; oracle/mpn_oracle.py: drop_stream)
__host__ __device__ inline unsigned drop_stream(unsigned base, int li) {
    if (li == 0) return base;
    if (base == kDropEncNode1 && li == 1) return kDropEncNode2;
    return base + 256u * (unsigned)li;
}

// ---- kernels --------------------------------------------------------------------------------------------------------------
// Wt[k][OP] = W[o][k] (zero padded), bp[OP] = b[o]: the operand form of gen_dense_kernel
__global__ __launch_bounds__(256) void tr_transpose_pad_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                               float* __restrict__ Wt, float* __restrict__ bp, int O, int K, int OP) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < (long long)K * OP) {
        const int k = (int)(t / OP), o = (int)(t - (long long)k * OP);
        Wt[t] = o < O ? W[(size_t)o * K + k] : 0.f;
    }
    if (t < OP) bp[t] = t < O ? b[t] : 0.f;
}

// out[r][:] = cat(seg0[r or idx0[r]], seg1[...], seg2[...])
__global__ __launch_bounds__(256) void tr_cat_kernel(GenSeg s0, GenSeg s1, GenSeg s2, float* __restrict__ out, long long M, int W) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * W) return;
    const long long r = t / W;
    int c = (int)(t - r * W);
    const GenSeg* sg = &s0;
    if (c >= s0.width) {
        c -= s0.width;
        sg = &s1;
        if (c >= s1.width) {
            c -= s1.width;
            sg = &s2;
        }
    }
    const long long rr = sg->idx ? (long long)sg->idx[r] : r;
    out[t] = sg->ptr[(size_t)rr * sg->ld + c];
}

// column reductions in double (torch's CPU BatchNorm accumulates in double): a block takes 64 columns x a chunk of rows
//   mode 0: out[0][c] += sum z                       mode 1: out[0][c] += sum (z - mean[c])^2
//   mode 2: out[0][c] += sum g,  out[1][c] += sum g * xhat   with xhat = (z - mean[c]) * invstd[c]
__global__ __launch_bounds__(256) void tr_col_reduce_kernel(const float* __restrict__ A, const float* __restrict__ Z,
                                                            const float* __restrict__ stat, long long M, int O, int rows_per_block,
                                                            double* __restrict__ out, int mode, const double* __restrict__ mean_sums) {
    __shared__ double s0[4][64], s1[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, M);
    double a0 = 0.0, a1 = 0.0;
    if (c < O) {
        // (mode 1 with `mean_sums`: the mean straight from the first pass's column sums -- tr_bn_mean_kernel's expression, same bits)
        const float mean = mode >= 1 ? (mean_sums ? (float)(mean_sums[c] / (double)M) : stat[c]) : 0.f, invstd = mode == 2 ? stat[O + c] : 0.f;
        for (long long r = r0 + ty; r < r1; r += 4) {
            const float z = Z[(size_t)r * O + c];
            if (mode == 0)
                a0 += (double)z;
            else if (mode == 1) {
                const double dz = (double)z - (double)mean;
                a0 += dz * dz;
            } else {
                const float g = A[(size_t)r * O + c];
                a0 += (double)g;
                a1 += (double)g * (double)((z - mean) * invstd);
            }
        }
    }
    s0[ty][tx] = a0, s1[ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && c < O) {
        a0 = ((s0[0][tx] + s0[1][tx]) + s0[2][tx]) + s0[3][tx];
        atomicAdd(&out[c], a0);
        if (mode == 2) {
            a1 = ((s1[0][tx] + s1[1][tx]) + s1[2][tx]) + s1[3][tx];
            atomicAdd(&out[O + c], a1);
        }
    }
}

// The same reductions for NARROW matrices (O <= 256: every MLP of the shipped shapes), where a 64-column tile would leave most lanes
// idle (O = 6: 6 of 64): the block walks its rows in passes of RB = 256 / O whole rows -- 256 consecutive floats, coalesced -- and a
// thread keeps ONE column (t % O) for all passes; eight passes are requested before they are added.
__global__ __launch_bounds__(256) void tr_col_reduce_narrow_kernel(const float* __restrict__ A, const float* __restrict__ Z,
                                                                   const float* __restrict__ stat, long long M, int O, int rows_per_block,
                                                                   double* __restrict__ out, int mode, const double* __restrict__ mean_sums) {
    __shared__ double s0[256], s1[256];
    const int RB = 256 / O;                       // rows per pass
    const int t = threadIdx.x;
    const int rl = t / O, c = t - rl * O;
    const bool on = rl < RB;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    double a0 = 0.0, a1 = 0.0;
    if (on) {
        const float mean = mode >= 1 ? (mean_sums ? (float)(mean_sums[c] / (double)M) : stat[c]) : 0.f, invstd = mode == 2 ? stat[O + c] : 0.f;
        for (long long rb = r0 + rl; rb < r1; rb += 8LL * RB) {
            float z[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long r = rb + (long long)u * RB;
                const bool ok = r < r1;
                z[u] = ok ? Z[(size_t)r * O + c] : mean;
                g[u] = (ok && mode == 2) ? A[(size_t)r * O + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = rb + (long long)u * RB < r1;
                if (mode == 0)
                    a0 += ok ? (double)z[u] : 0.0;
                else if (mode == 1) {
                    const double dz = (double)z[u] - (double)mean;
                    a0 += dz * dz;
                } else {
                    a0 += (double)g[u];
                    a1 += (double)g[u] * (double)((z[u] - mean) * invstd);
                }
            }
        }
    }
    s0[t] = a0, s1[t] = a1;
    __syncthreads();
    if (t < O) {   // the RB row lanes of column t, in order
        double b0 = 0.0, b1 = 0.0;
        for (int q = 0; q < RB; ++q) b0 += s0[q * O + t], b1 += s1[q * O + t];
        atomicAdd(&out[t], b0);
        if (mode == 2) atomicAdd(&out[O + t], b1);
    }
}

__global__ void tr_bn_mean_kernel(const double* __restrict__ sums, long long M, int O, float* __restrict__ stat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < O) stat[c] = (float)(sums[c] / (double)M);
}

// invstd from the biased variance; running buffers as torch.nn.BatchNorm1d updates them (momentum 0.1, unbiased variance)
__global__ void tr_bn_var_kernel(const double* __restrict__ sums, long long M, int O, float* __restrict__ stat,
                                 float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= O) return;
    const double var = sums[c] / (double)M;
    stat[O + c] = (float)(1.0 / sqrt(var + (double)kBnEps));
    const double unb = M > 1 ? sums[c] / (double)(M - 1) : var;
    running_mean[c] = (float)((double)kBnMomentum * (double)stat[c] + (1.0 - (double)kBnMomentum) * (double)running_mean[c]);
    running_var[c] = (float)((double)kBnMomentum * unb + (1.0 - (double)kBnMomentum) * (double)running_var[c]);
}

// a = [ReLU](z * alpha + beta'), alpha = invstd * gamma, beta' = beta - mean * alpha (the form of torch's CPU kernel)
// (z and a may be the same buffer: the stand-alone MLP evaluation normalises in place)
__global__ __launch_bounds__(256) void tr_bn_apply_kernel(const float* z, const float* __restrict__ stat,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* a, long long M, int O, int relu) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * O) return;
    const int c = (int)(t % O);
    const float alpha = stat[O + c] * gamma[c];
    const float b2 = beta[c] - stat[c] * alpha;
    const float y = z[t] * alpha + b2;
    a[t] = relu ? fmaxf(y, 0.f) : y;
}

// Train-mode BatchNorm forward behind the two column passes, in ONE launch (round 5; O <= 256): every block derives mean / invstd of the
// columns from the sums (tr_bn_mean_kernel's and tr_bn_var_kernel's expressions: same bits) into LDS and normalises its 256 elements; block
// 0 also stores them in `stat` (the backward reads them there) and updates the running buffers as torch.nn.BatchNorm1d does.
// sums[0 .. O) = sum z, sums[O .. 2 O) = sum (z - mean)^2.
__global__ __launch_bounds__(256) void tr_bn_apply_fused_kernel(const float* z, const double* __restrict__ sums, float* __restrict__ stat,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float* a, long long M, int O, int relu) {
    __shared__ float s_alpha[256], s_b2[256];
    const int tid = threadIdx.x;
    if (tid < O) {
        const float mean = (float)(sums[tid] / (double)M);
        const double var = sums[O + tid] / (double)M;
        const float invstd = (float)(1.0 / sqrt(var + (double)kBnEps));
        const float alpha = invstd * gamma[tid];
        s_alpha[tid] = alpha;
        s_b2[tid] = beta[tid] - mean * alpha;
        if (blockIdx.x == 0) {
            stat[tid] = mean;
            stat[O + tid] = invstd;
            const double unb = M > 1 ? sums[O + tid] / (double)(M - 1) : var;
            running_mean[tid] = (float)((double)kBnMomentum * (double)mean + (1.0 - (double)kBnMomentum) * (double)running_mean[tid]);
            running_var[tid] = (float)((double)kBnMomentum * unb + (1.0 - (double)kBnMomentum) * (double)running_var[tid]);
        }
    }
    __syncthreads();
    const long long t = (long long)blockIdx.x * 256 + tid;
    if (t >= M * O) return;
    const int c = (int)(t % O);
    const float y = z[t] * s_alpha[c] + s_b2[c];
    a[t] = relu ? fmaxf(y, 0.f) : y;
}

// g <- d z = gamma * invstd * (g - s1 / M - xhat * s2 / M);  d gamma += s2, d beta += s1
__global__ __launch_bounds__(256) void tr_bn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ z,
                                                              const float* __restrict__ stat, const float* __restrict__ gamma,
                                                              const double* __restrict__ sums, long long M, int O,
                                                              float* __restrict__ d_gamma, float* __restrict__ d_beta) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < O) {
        if (d_gamma) d_gamma[t] += (float)sums[O + t];
        if (d_beta) d_beta[t] += (float)sums[t];
    }
    if (t >= M * O) return;
    const int c = (int)(t % O);
    const float invstd = stat[O + c];
    const float xhat = (z[t] - stat[c]) * invstd;
    const float s1 = (float)(sums[c] / (double)M), s2 = (float)(sums[O + c] / (double)M);
    g[t] = gamma[c] * invstd * (g[t] - s1 - xhat * s2);
}

// out[r][k] = sum_o G[r][o] * W[o][k]
__global__ __launch_bounds__(256) void tr_matmul_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                        float* __restrict__ out, long long M, int O, int K) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * K) return;
    const long long r = t / K;
    const int k = (int)(t - r * K);
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(G[(size_t)r * O + o], W[(size_t)o * K + k], acc);
    out[t] = acc;
}

// dst[idx[r] or r][dst_off + c] += src[r][src_off + c]   (atomics when gathered: several rows share a destination)
__global__ __launch_bounds__(256) void tr_scatter_add_kernel(const float* __restrict__ src, int ld_src, int src_off, int width,
                                                             const int* __restrict__ idx, float* __restrict__ dst, int ld_dst,
                                                             int dst_off, long long M) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * width) return;
    const long long r = t / width;
    const int c = (int)(t - r * width);
    const float v = src[(size_t)r * ld_src + src_off + c];
    if (idx)
        atomicAdd(&dst[(size_t)idx[r] * ld_dst + dst_off + c], v);
    else
        dst[(size_t)r * ld_dst + dst_off + c] += v;
}

// dst[r][c] = src[r][src_off + c]
__global__ __launch_bounds__(256) void tr_slice_kernel(const float* __restrict__ src, int ld_src, int src_off, int width,
                                                       float* __restrict__ dst, long long M) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * width) return;
    const long long r = t / width;
    dst[t] = src[(size_t)r * ld_src + src_off + (int)(t - r * width)];
}

// arg[i][c] = the lowest edge id k (caller's order) with row[k] == i and m[k][c] == h[i][c]   (arg pre-set to INT_MAX)
__global__ __launch_bounds__(256) void tr_max_arg_kernel(const float* __restrict__ m, const float* __restrict__ h,
                                                         const int* __restrict__ row, int* __restrict__ arg, long long E, int H) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= E * H) return;
    const long long k = t / H;
    const int c = (int)(t - k * H);
    const int i = row[k];
    if (m[t] == h[(size_t)i * H + c]) atomicMin(&arg[(size_t)i * H + c], (int)k);
}

// d m[k][c] from d h[row[k]][c]: sum -> copy, mean -> / degree, max -> only the arg edge
__global__ __launch_bounds__(256) void tr_agg_bwd_kernel(const float* __restrict__ dh, const int* __restrict__ row,
                                                         const int* __restrict__ seg_ptr, const int* __restrict__ arg,
                                                         float* __restrict__ dm, long long E, int H, int agg) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= E * H) return;
    const long long k = t / H;
    const int c = (int)(t - k * H);
    const int i = row[k];
    float g = dh[(size_t)i * H + c];
    if (agg == GNNCCA_AGG_MEAN) g = g / (float)max(seg_ptr[i + 1] - seg_ptr[i], 1);
    if (agg == GNNCCA_AGG_MAX && arg[(size_t)i * H + c] != (int)k) g = 0.f;
    dm[t] = g;
}

// ---- the tape ----------------------------------------------------------------------------------------------------------------
struct TrLayer {
    size_t z, a, stat;   // byte offsets: Linear output (BatchNorm layers only), layer output, [2][out] batch mean | invstd
};
struct TrCall {          // one application of an MLP (models/mlp.py:26-28)
    int mlp;             // 0 encoder.node, 1 encoder.edge, 2 MPNet.edge_model, 3 MPNet.node_model, 4 classifier.edge
    long long M;
    size_t xin;          // materialised input [M][in] (cat), or SIZE_MAX: the input lives elsewhere (x, edge_attr, a layer output)
    TrLayer lay[GNNCCA_MAX_LAYERS];
    unsigned drop_base;
    float p;
};
struct TrPlan {
    size_t flags, blockflags, seg_ptr, col32, perm, cursor, row32, colo32;
    size_t wt[5][GNNCCA_MAX_LAYERS], bp[5][GNNCCA_MAX_LAYERS];
    int p0[5];                       // index of each MLP's first entry in the parameter list
    TrCall enc_node, enc_edge;
    std::vector<TrCall> edge, node, cls;   // per step; per classified step
    std::vector<size_t> h, hin, ein, arg;  // per step: aggregated node latents [N][H]; cat(h0, h) / cat(e0, e) when reattaching; max arg
    size_t dsum;                     // [2][max width] doubles
    size_t dh, dhin, dh0, de, de0, gE[2], gN[2];   // backward scratch
    int hin_w, ein_w, max_we, max_wn;
    size_t total;
};

static size_t tr_out_of(const gnncca_mlp& m, int fallback) { return m.n_layers > 0 ? (size_t)m.layers[m.n_layers - 1].out_dim : (size_t)fallback; }

static bool tr_plan(const gnncca_mpn_dims* d, int64_t n, int64_t e, TrPlan* P) {
    if (!dims_valid(d)) return false;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t N = (size_t)std::max<int64_t>(n, 0), E = (size_t)std::max<int64_t>(e, 0);
    const int H = d->node_dim, EF = d->edge_dim;
    P->flags = take(256);
    P->blockflags = take((E / 256 + 2) * 4);
    P->seg_ptr = take((N + 1) * 4);
    P->col32 = take(E * 4);
    P->perm = take(E * 4);
    P->cursor = take((N + 1) * 4);
    P->row32 = take(E * 4);
    P->colo32 = take(E * 4);
    int pidx = 0, maxw = 1;
    for (int m = 0; m < 5; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        P->p0[m] = pidx;
        pidx += mlp_param_count(mlp);
        for (int l = 0; l < mlp.n_layers; ++l) {
            const int OP = (mlp.layers[l].out_dim + 7) / 8 * 8;
            P->wt[m][l] = take((size_t)mlp.layers[l].in_dim * OP * 4);
            P->bp[m][l] = take((size_t)OP * 4);
            maxw = std::max(maxw, (int)mlp.layers[l].out_dim);
        }
    }
    P->hin_w = d->reattach_nodes ? 2 * H : H;
    P->ein_w = d->reattach_edges ? 2 * EF : EF;
    auto make_call = [&](int m, long long M, int in_w, bool materialise, unsigned drop_base) {
        TrCall c;
        std::memset(&c, 0, sizeof(c));
        c.mlp = m;
        c.M = M;
        c.drop_base = drop_base;
        c.xin = materialise ? take((size_t)M * in_w * 4) : SIZE_MAX;
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) {
            const size_t O = (size_t)mlp.layers[l].out_dim;
            c.lay[l].z = mlp.layers[l].has_bn ? take((size_t)M * O * 4) : SIZE_MAX;
            c.lay[l].a = take((size_t)M * O * 4);
            c.lay[l].stat = mlp.layers[l].has_bn ? take(2 * O * 4) : SIZE_MAX;
        }
        return c;
    };
    P->enc_node = make_call(0, (long long)N, d->node_in, false, kDropEncNode1);
    P->enc_edge = make_call(1, (long long)E, d->edge_in, false, kDropEncEdge);
    P->edge.clear(), P->node.clear(), P->cls.clear(), P->h.clear(), P->hin.clear(), P->ein.clear(), P->arg.clear();
    const int L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    for (int step = 1; step <= L; ++step) {
        P->hin.push_back(d->reattach_nodes ? take(N * (size_t)P->hin_w * 4) : SIZE_MAX);
        P->ein.push_back(d->reattach_edges ? take(E * (size_t)P->ein_w * 4) : SIZE_MAX);
        P->edge.push_back(make_call(2, (long long)E, 2 * P->hin_w + P->ein_w, true, kDropEdgeStep + (unsigned)step));
        P->node.push_back(make_call(3, (long long)E, P->hin_w + EF, true, kDropNodeStep + (unsigned)step));
        P->h.push_back(take(N * (size_t)H * 4));
        P->arg.push_back(d->agg == GNNCCA_AGG_MAX ? take(N * (size_t)H * 4) : SIZE_MAX);
        if (step >= first_cls) P->cls.push_back(make_call(4, (long long)E, EF, false, kDropCls + (unsigned)P->cls.size()));
    }
    if (L == 0) P->cls.push_back(make_call(4, (long long)E, EF, false, kDropCls));
    // backward scratch
    int we = std::max(std::max(2 * P->hin_w + P->ein_w, P->hin_w + EF), std::max(H, EF));
    for (int m = 1; m <= 4; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) we = std::max(we, (int)std::max(mlp.layers[l].out_dim, m == 1 ? 1 : mlp.layers[l].in_dim));
    }
    int wn = std::max(P->hin_w, H);
    for (int l = 0; l < d->enc_node.n_layers; ++l) wn = std::max(wn, (int)d->enc_node.layers[l].out_dim);
    P->max_we = we, P->max_wn = wn;
    P->dsum = take(2 * (size_t)std::max(maxw, 1) * 8);
    P->dh = take(N * (size_t)H * 4);
    P->dhin = take(N * (size_t)P->hin_w * 4);
    P->dh0 = take(N * (size_t)H * 4);
    P->de = take(E * (size_t)EF * 4);
    P->de0 = take(E * (size_t)EF * 4);
    for (int i = 0; i < 2; ++i) P->gE[i] = take(E * (size_t)we * 4);
    for (int i = 0; i < 2; ++i) P->gN[i] = take(N * (size_t)wn * 4);
    P->total = off;
    return true;
}

struct TrCtx {
    const gnncca_mpn_dims* d;
    const TrPlan* P;
    char* base;
    float* const* params;
    float* const* grads;   // backward only
    DropCfg drop;
    hipStream_t st;
    float* at(size_t off) const { return reinterpret_cast<float*>(base + off); }
    float p_of(int mlp) const { return mlp <= 1 ? drop.p_enc : (mlp == 2 ? drop.p_edge : (mlp == 3 ? drop.p_node : drop.p_cls)); }
    // index of layer l's weight in the parameter list (weight, bias, [BatchNorm weight, bias, running_mean, running_var])
    int pw(int mlp, int l) const {
        const gnncca_mlp& m = mlp_by_index(d, mlp);
        int i = P->p0[mlp];
        for (int q = 0; q < l; ++q) i += 2 + (m.layers[q].has_bn ? 4 : 0);
        return i;
    }
};

// `second_pass` (mode 1 only): the squared deviations go to sums[O .. 2 O) next to the first pass's column sums, which stay (no memset) and
// supply the mean -- the fused train-mode BatchNorm forward reads both
static int tr_col_reduce(const TrCtx& c, const float* A, const float* Z, const float* stat, long long M, int O, int mode, bool second_pass = false) {
    double* sums = reinterpret_cast<double*>(c.base + c.P->dsum);
    const double* mean_sums = second_pass ? sums : nullptr;
    if (second_pass)
        sums += O;
    else
        HIP_TRY(hipMemsetAsync(sums, 0, 2 * (size_t)O * 8, c.st));
    if (O <= 256) {
        const int RB = 256 / O;
        long long rows = (M + 1023) / 1024;                       // aim at ~1024 blocks ...
        rows = std::max<long long>((rows + 8LL * RB - 1) / (8LL * RB) * (8LL * RB), 8LL * RB);   // ... of whole rounds of 8 passes
        hipLaunchKernelGGL(tr_col_reduce_narrow_kernel, dim3((unsigned)((M + rows - 1) / rows)), dim3(256), 0, c.st, A, Z, stat, M, O,
                           (int)rows, sums, mode, mean_sums);
    } else {
        const int rows = 1024;
        hipLaunchKernelGGL(tr_col_reduce_kernel, dim3((unsigned)((O + 63) / 64), (unsigned)((M + rows - 1) / rows)), dim3(256), 0, c.st, A, Z,
                           stat, M, O, rows, sums, mode, mean_sums);
    }
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

// forward of one MLP call; returns the pointer to its output through *out
static int tr_mlp_forward(const TrCtx& c, const TrCall& call, const float* xin, const float** out) {
    const gnncca_mlp& mlp = mlp_by_index(c.d, call.mlp);
    const GenSeg none = {nullptr, nullptr, 0, 0};
    const float* cur = xin;
    const long long M = call.M;
    const float p = c.p_of(call.mlp);
    for (int l = 0; l < mlp.n_layers; ++l) {
        const gnncca_layer& L = mlp.layers[l];
        const int O = L.out_dim, K = L.in_dim, OP = (O + 7) / 8 * 8;
        const int pi = c.pw(call.mlp, l);
        float* dst = L.has_bn ? c.at(call.lay[l].z) : c.at(call.lay[l].a);
        if (M * O > 0) {
            for (int o0 = 0; o0 < O; o0 += 2048) {
                const int og = std::min(2048, O - o0);
                const int rows_per_block = 256 / ((og + 7) / 8);
                hipLaunchKernelGGL(gen_dense_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, c.st,
                                   GenSeg{cur, nullptr, K, K}, none, none, (const float*)c.at(c.P->wt[call.mlp][l]) + o0,
                                   (const float*)c.at(c.P->bp[call.mlp][l]) + o0, dst + o0, M, K, og, OP, O, (L.has_bn ? 0 : L.relu));
            }
            HIP_TRY(hipGetLastError());
            if (L.has_bn) {
                float* stat = c.at(call.lay[l].stat);
                const double* sums = reinterpret_cast<const double*>(c.base + c.P->dsum);
                int s = tr_col_reduce(c, nullptr, dst, nullptr, M, O, 0);
                if (s != GNNCCA_OK) return s;
                if (O <= 256) {
                    // round 5: 8 -> 5 operations per train-mode BatchNorm layer (the mean and the variance no longer have kernels of their own;
                    // one memset for both passes' sums; the statistics, the running buffers and the normalisation in one launch) -- same bits
                    s = tr_col_reduce(c, nullptr, dst, nullptr, M, O, 1, true);
                    if (s != GNNCCA_OK) return s;
                    hipLaunchKernelGGL(tr_bn_apply_fused_kernel, grid1((size_t)M * O, 256), dim3(256), 0, c.st, (const float*)dst, sums, stat,
                                       (const float*)c.params[pi + 2], (const float*)c.params[pi + 3], c.params[pi + 4], c.params[pi + 5],
                                       c.at(call.lay[l].a), M, O, (int)L.relu);
                } else {
                    hipLaunchKernelGGL(tr_bn_mean_kernel, dim3((O + 255) / 256), dim3(256), 0, c.st, sums, M, O, stat);
                    s = tr_col_reduce(c, nullptr, dst, stat, M, O, 1);
                    if (s != GNNCCA_OK) return s;
                    hipLaunchKernelGGL(tr_bn_var_kernel, dim3((O + 255) / 256), dim3(256), 0, c.st, sums, M, O, stat, c.params[pi + 4],
                                       c.params[pi + 5]);
                    hipLaunchKernelGGL(tr_bn_apply_kernel, grid1((size_t)M * O, 256), dim3(256), 0, c.st, (const float*)dst,
                                       (const float*)stat, (const float*)c.params[pi + 2], (const float*)c.params[pi + 3],
                                       c.at(call.lay[l].a), M, O, (int)L.relu);
                }
                HIP_TRY(hipGetLastError());
            }
            if (L.relu && p > 0.f) {
                hipLaunchKernelGGL(apply_dropout_kernel, grid1((size_t)M * O, 256), dim3(256), 0, c.st, c.at(call.lay[l].a),
                                   (long long)M * O, c.drop, drop_stream(call.drop_base, l), p);
                HIP_TRY(hipGetLastError());
            }
        }
        cur = c.at(call.lay[l].a);
    }
    *out = cur;
    return GNNCCA_OK;
}

// backward of one MLP call.  `g` = d loss / d output ([M][out], overwritten); `gbuf` two scratch buffers of at least M x the
// widest layer; d input goes to *dx ([M][in], one of the scratch buffers or g itself for an empty MLP) when want_dx.
static int tr_mlp_backward(const TrCtx& c, const TrCall& call, const float* xin, float* g, float* gbuf0, float* gbuf1, bool want_dx,
                           float** dx) {
    const gnncca_mlp& mlp = mlp_by_index(c.d, call.mlp);
    const long long M = call.M;
    const float p = c.p_of(call.mlp);
    float* cur = g;
    for (int l = mlp.n_layers - 1; l >= 0; --l) {
        const gnncca_layer& L = mlp.layers[l];
        const int O = L.out_dim, K = L.in_dim;
        const int pi = c.pw(call.mlp, l);
        if (M * O > 0) {
            if (L.relu) {
                hipLaunchKernelGGL(bwd_relu_mask_kernel, grid1((size_t)M * O, 256), dim3(256), 0, c.st, cur,
                                   (const float*)c.at(call.lay[l].a), (long long)M * O, p > 0.f ? 1.f / (1.f - p) : 1.f);
                HIP_TRY(hipGetLastError());
            }
            if (L.has_bn) {
                const float* z = c.at(call.lay[l].z);
                const float* stat = c.at(call.lay[l].stat);
                int s = tr_col_reduce(c, cur, z, stat, M, O, 2);
                if (s != GNNCCA_OK) return s;
                hipLaunchKernelGGL(tr_bn_bwd_apply_kernel, grid1(std::max<size_t>((size_t)M * O, (size_t)O), 256), dim3(256), 0, c.st, cur, z,
                                   stat, (const float*)c.params[pi + 2], reinterpret_cast<const double*>(c.base + c.P->dsum), M, O,
                                   c.grads[pi + 2], c.grads[pi + 3]);
                HIP_TRY(hipGetLastError());
            }
            const float* X = l == 0 ? xin : c.at(call.lay[l - 1].a);
            if (c.grads[pi] != nullptr || c.grads[pi + 1] != nullptr) {
                // d W[o][k] += sum_r g[r][o] X[r][k], d b[o] += sum_r g[r][o]
                HIP_TRY(launch_outer(cur, O, X, K, c.grads[pi], K, c.grads[pi + 1], (int)M, O, K, c.st));
            }
            if (l > 0 || want_dx) {
                float* nxt = cur == gbuf0 ? gbuf1 : gbuf0;
                hipLaunchKernelGGL(tr_matmul_kernel, grid1((size_t)M * K, 256), dim3(256), 0, c.st, (const float*)cur,
                                   (const float*)c.params[pi], nxt, M, O, K);
                HIP_TRY(hipGetLastError());
                cur = nxt;
            }
        }
    }
    if (dx) *dx = cur;
    return GNNCCA_OK;
}

static int tr_prepare(const gnncca_mpn_dims* d, float* const* params, int n_params, int64_t n_nodes, int64_t n_edges, void* tape,
                      size_t tape_bytes, TrPlan* P) {
    if (!d || !params || !tape) return GNNCCA_ERR_INVALID_ARG;
    if (n_nodes < 0 || n_edges < 0 || n_nodes > 0x7fffffffLL || n_edges > 0x7fffffffLL) return GNNCCA_ERR_INVALID_ARG;
    if (!tr_plan(d, n_nodes, n_edges, P)) return GNNCCA_ERR_INVALID_ARG;
    if (n_params != gnncca_param_count(d)) return GNNCCA_ERR_INVALID_ARG;
    if (tape_bytes < P->total) return GNNCCA_ERR_WORKSPACE;
    return GNNCCA_OK;
}

static DropCfg tr_dropcfg(const gnncca_dropout* dropout) {
    DropCfg dc;
    std::memset(&dc, 0, sizeof(dc));
    if (dropout && dropout->seed_dev) {
        dc.p_enc = dropout->p_enc, dc.p_edge = dropout->p_edge, dc.p_node = dropout->p_node, dc.p_cls = dropout->p_cls;
        dc.seed = reinterpret_cast<const unsigned long long*>(dropout->seed_dev);
    }
    return dc;
}

static int train_forward_impl(const gnncca_mpn_dims* d, float* const* params, int n_params, const float* x, const int64_t* edge_index,
                              const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* tape, size_t tape_bytes,
                              float* logits_out, const gnncca_dropout* dropout, hipStream_t st) {
    TrPlan P;
    int s = tr_prepare(d, params, n_params, n_nodes, n_edges, tape, tape_bytes, &P);
    if (s != GNNCCA_OK) return s;
    TrCtx c;
    c.d = d, c.P = &P, c.base = static_cast<char*>(tape), c.params = params, c.grads = nullptr, c.drop = tr_dropcfg(dropout), c.st = st;
    const int N = (int)n_nodes, E = (int)n_edges, H = d->node_dim, EF = d->edge_dim;
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    unsigned* flags = reinterpret_cast<unsigned*>(c.base + P.flags);
    unsigned* blockflags = reinterpret_cast<unsigned*>(c.base + P.blockflags);
    int* seg_ptr = reinterpret_cast<int*>(c.base + P.seg_ptr);
    int* col32 = reinterpret_cast<int*>(c.base + P.col32);
    int* perm = reinterpret_cast<int*>(c.base + P.perm);
    int* cursor = reinterpret_cast<int*>(c.base + P.cursor);
    int* row32 = reinterpret_cast<int*>(c.base + P.row32);
    int* colo32 = reinterpret_cast<int*>(c.base + P.colo32);
    // graph plan (CSR offsets by source node, caller-order 32-bit indices, flag word): the inference path's kernels
    if (E > 0) {
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.ei = ei, ep.seg_ptr = seg_ptr, ep.col32 = col32, ep.blockflags = blockflags, ep.E = E, ep.N = N;
        hipLaunchKernelGGL(plan_only_kernel, dim3(plan_num_blocks(E)), dim3(256), 0, st, ep);
        hipLaunchKernelGGL(gen_index32_kernel, dim3((E + 255) / 256), dim3(256), 0, st, ei, E, N, row32, colo32);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(gen_plan_finish_kernel, dim3(1), dim3(256), 0, st, ei, E, N, seg_ptr, col32, perm, cursor, flags,
                       (const unsigned*)blockflags);
    HIP_TRY(hipGetLastError());
    // operand form of the weights (this iteration's values)
    for (int m = 0; m < 5; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) {
            const int O = mlp.layers[l].out_dim, K = mlp.layers[l].in_dim, OP = (O + 7) / 8 * 8, pi = c.pw(m, l);
            hipLaunchKernelGGL(tr_transpose_pad_kernel, grid1((size_t)K * OP, 256), dim3(256), 0, st, (const float*)params[pi],
                               (const float*)params[pi + 1], c.at(P.wt[m][l]), c.at(P.bp[m][l]), O, K, OP);
        }
    }
    HIP_TRY(hipGetLastError());
    // encoder (models/mpn.py:270)
    const float *h0 = x, *e0 = edge_attr;
    s = tr_mlp_forward(c, P.enc_node, x, &h0);
    if (s != GNNCCA_OK) return s;
    if (E == 0) return GNNCCA_OK;
    s = tr_mlp_forward(c, P.enc_edge, edge_attr, &e0);
    if (s != GNNCCA_OK) return s;
    const int L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    const GenSeg none = {nullptr, nullptr, 0, 0};
    int out_idx = 0;
    auto classify = [&](const float* ee) -> int {
        const float* lo = nullptr;
        int r = tr_mlp_forward(c, P.cls[out_idx], ee, &lo);
        if (r != GNNCCA_OK) return r;
        float* dst = logits_out + (size_t)out_idx * E;
        HIP_TRY(hipMemcpyAsync(dst, lo, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(gen_poison_kernel, grid1((size_t)E, 256), dim3(256), 0, st, dst, (long long)E, (const unsigned*)flags);
        HIP_TRY(hipGetLastError());
        ++out_idx;
        return GNNCCA_OK;
    };
    if (L == 0) return classify(e0);
    const float *h_cur = h0, *e_cur = e0;
    for (int step = 1; step <= L; ++step) {
        const int si = step - 1;
        const float* hin = h_cur;
        const float* ein = e_cur;
        if (d->reattach_nodes) {   // models/mpn.py:285: cat(initial, latent)
            hipLaunchKernelGGL(tr_cat_kernel, grid1((size_t)N * P.hin_w, 256), dim3(256), 0, st, GenSeg{h0, nullptr, H, H},
                               GenSeg{h_cur, nullptr, H, H}, none, c.at(P.hin[si]), (long long)N, P.hin_w);
            hin = c.at(P.hin[si]);
        }
        if (d->reattach_edges) {   // models/mpn.py:283
            hipLaunchKernelGGL(tr_cat_kernel, grid1((size_t)E * P.ein_w, 256), dim3(256), 0, st, GenSeg{e0, nullptr, EF, EF},
                               GenSeg{e_cur, nullptr, EF, EF}, none, c.at(P.ein[si]), (long long)E, P.ein_w);
            ein = c.at(P.ein[si]);
        }
        // edge update (models/mpn.py:48,68-69): cat(x[row], x[col], e)
        const int we = 2 * P.hin_w + P.ein_w;
        hipLaunchKernelGGL(tr_cat_kernel, grid1((size_t)E * we, 256), dim3(256), 0, st, GenSeg{hin, row32, P.hin_w, P.hin_w},
                           GenSeg{hin, colo32, P.hin_w, P.hin_w}, GenSeg{ein, nullptr, P.ein_w, P.ein_w}, c.at(P.edge[si].xin),
                           (long long)E, we);
        HIP_TRY(hipGetLastError());
        const float* e_new = nullptr;
        s = tr_mlp_forward(c, P.edge[si], c.at(P.edge[si].xin), &e_new);
        if (s != GNNCCA_OK) return s;
        // node update (models/mpn.py:97-99): cat(x[row], e'), aggregate by row
        const int wn = P.hin_w + EF;
        hipLaunchKernelGGL(tr_cat_kernel, grid1((size_t)E * wn, 256), dim3(256), 0, st, GenSeg{hin, row32, P.hin_w, P.hin_w},
                           GenSeg{e_new, nullptr, EF, EF}, none, c.at(P.node[si].xin), (long long)E, wn);
        HIP_TRY(hipGetLastError());
        const float* msg = nullptr;
        s = tr_mlp_forward(c, P.node[si], c.at(P.node[si].xin), &msg);
        if (s != GNNCCA_OK) return s;
        hipLaunchKernelGGL(gen_aggregate_kernel, dim3((unsigned)N), dim3(256), 0, st, msg, (const int*)seg_ptr, (const int*)perm,
                           (const unsigned*)flags, c.at(P.h[si]), N, H, (int)d->agg);
        HIP_TRY(hipGetLastError());
        h_cur = c.at(P.h[si]);
        e_cur = e_new;
        if (step >= first_cls) {
            s = classify(e_cur);
            if (s != GNNCCA_OK) return s;
        }
    }
    return GNNCCA_OK;
}

static int train_backward_impl(const gnncca_mpn_dims* d, float* const* params, int n_params, const float* x, const int64_t* edge_index,
                               const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* tape, size_t tape_bytes,
                               const float* grad_logits, float* const* grads, const gnncca_dropout* dropout, hipStream_t st) {
    TrPlan P;
    int s = tr_prepare(d, params, n_params, n_nodes, n_edges, tape, tape_bytes, &P);
    if (s != GNNCCA_OK) return s;
    if (!grads) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges == 0) return GNNCCA_OK;   // no edge, no logit: nothing reaches any parameter (an empty grad_logits may be a null pointer)
    if (!grad_logits) return GNNCCA_ERR_INVALID_ARG;
    TrCtx c;
    c.d = d, c.P = &P, c.base = static_cast<char*>(tape), c.params = params, c.grads = grads, c.drop = tr_dropcfg(dropout), c.st = st;
    const int N = (int)n_nodes, E = (int)n_edges, H = d->node_dim, EF = d->edge_dim;
    const int* seg_ptr = reinterpret_cast<const int*>(c.base + P.seg_ptr);
    const int* row32 = reinterpret_cast<const int*>(c.base + P.row32);
    const int* colo32 = reinterpret_cast<const int*>(c.base + P.colo32);
    const gnncca_mlp& cls = d->cls_edge;
    float *dh = c.at(P.dh), *dhin = c.at(P.dhin), *dh0 = c.at(P.dh0), *de = c.at(P.de), *de0 = c.at(P.de0);
    float *gE0 = c.at(P.gE[0]), *gE1 = c.at(P.gE[1]), *gN0 = c.at(P.gN[0]), *gN1 = c.at(P.gN[1]);
    HIP_TRY(hipMemsetAsync(dh, 0, (size_t)N * H * 4, st));
    HIP_TRY(hipMemsetAsync(dh0, 0, (size_t)N * H * 4, st));
    HIP_TRY(hipMemsetAsync(de, 0, (size_t)E * EF * 4, st));
    HIP_TRY(hipMemsetAsync(de0, 0, (size_t)E * EF * 4, st));
    const float* h0 = d->enc_node.n_layers > 0 ? c.at(P.enc_node.lay[d->enc_node.n_layers - 1].a) : x;
    const float* e0 = d->enc_edge.n_layers > 0 ? c.at(P.enc_edge.lay[d->enc_edge.n_layers - 1].a) : edge_attr;
    const int L = d->num_enc_steps, first_cls = L - d->num_class_steps + 1;
    auto out_of = [&](const TrCall& call, const float* in_if_empty) -> const float* {
        const gnncca_mlp& m = mlp_by_index(d, call.mlp);
        return m.n_layers > 0 ? c.at(call.lay[m.n_layers - 1].a) : in_if_empty;
    };
    // d loss / d (classifier input) added to `de`
    auto classify_bwd = [&](int idx, const float* ee) -> int {
        float* g = gE0;   // the classifier's output gradient is the caller's: work on a copy
        HIP_TRY(hipMemcpyAsync(g, grad_logits + (size_t)idx * E, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        float* dx = nullptr;
        int r = tr_mlp_backward(c, P.cls[idx], ee, g, gE0, gE1, true, &dx);
        if (r != GNNCCA_OK) return r;
        hipLaunchKernelGGL(bwd_add_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, de, (const float*)dx, (long long)E * EF);
        HIP_TRY(hipGetLastError());
        return GNNCCA_OK;
    };
    if (L == 0) {
        if (cls.n_layers == 0) return GNNCCA_ERR_UNSUPPORTED;
        s = classify_bwd(0, e0);
        if (s != GNNCCA_OK) return s;
    }
    int idx = (int)P.cls.size() - 1;
    for (int step = L; step >= 1; --step) {
        const int si = step - 1;
        const float* h_prev = step == 1 ? h0 : c.at(P.h[si - 1]);
        (void)h_prev;
        const float* e_new = out_of(P.edge[si], nullptr);
        const float* msg = out_of(P.node[si], nullptr);
        if (step >= first_cls) {
            s = classify_bwd(idx--, e_new);
            if (s != GNNCCA_OK) return s;
        }
        // aggregator backward: d h' -> d messages
        float* dm = gE0;
        if (d->agg == GNNCCA_AGG_MAX) {
            int* arg = reinterpret_cast<int*>(c.base + P.arg[si]);
            HIP_TRY(hipMemsetAsync(arg, 0x7f, (size_t)N * H * 4, st));
            hipLaunchKernelGGL(tr_max_arg_kernel, grid1((size_t)E * H, 256), dim3(256), 0, st, msg, (const float*)c.at(P.h[si]), row32, arg,
                               (long long)E, H);
        }
        hipLaunchKernelGGL(tr_agg_bwd_kernel, grid1((size_t)E * H, 256), dim3(256), 0, st, (const float*)dh, row32, seg_ptr,
                           d->agg == GNNCCA_AGG_MAX ? reinterpret_cast<const int*>(c.base + P.arg[si]) : nullptr, dm, (long long)E, H,
                           (int)d->agg);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(dhin, 0, (size_t)N * P.hin_w * 4, st));
        // node model backward: d cat(x[row], e')
        float* dxn = nullptr;
        s = tr_mlp_backward(c, P.node[si], c.at(P.node[si].xin), dm, gE0, gE1, true, &dxn);
        if (s != GNNCCA_OK) return s;
        const int wn = P.hin_w + EF;
        hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)E * P.hin_w, 256), dim3(256), 0, st, (const float*)dxn, wn, 0, P.hin_w,
                           row32, dhin, P.hin_w, 0, (long long)E);
        hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, (const float*)dxn, wn, P.hin_w, EF,
                           (const int*)nullptr, de, EF, 0, (long long)E);
        HIP_TRY(hipGetLastError());
        // edge model backward: d cat(x[row], x[col], e); its output gradient `de` is consumed (copied: the MLP overwrites it)
        float* ge = gE0;
        HIP_TRY(hipMemcpyAsync(ge, de, (size_t)E * EF * 4, hipMemcpyDeviceToDevice, st));
        float* dxe = nullptr;
        s = tr_mlp_backward(c, P.edge[si], c.at(P.edge[si].xin), ge, gE0, gE1, true, &dxe);
        if (s != GNNCCA_OK) return s;
        const int we = 2 * P.hin_w + P.ein_w;
        hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)E * P.hin_w, 256), dim3(256), 0, st, (const float*)dxe, we, 0, P.hin_w,
                           row32, dhin, P.hin_w, 0, (long long)E);
        hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)E * P.hin_w, 256), dim3(256), 0, st, (const float*)dxe, we, P.hin_w,
                           P.hin_w, colo32, dhin, P.hin_w, 0, (long long)E);
        HIP_TRY(hipGetLastError());
        // split the gradients of the (possibly reattached) inputs into initial / latent parts
        if (d->reattach_edges) {
            hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, (const float*)dxe, we, 2 * P.hin_w, EF,
                               (const int*)nullptr, de0, EF, 0, (long long)E);
            hipLaunchKernelGGL(tr_slice_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, (const float*)dxe, we, 2 * P.hin_w + EF, EF,
                               de, (long long)E);
        } else {
            hipLaunchKernelGGL(tr_slice_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, (const float*)dxe, we, 2 * P.hin_w, EF, de,
                               (long long)E);
        }
        if (d->reattach_nodes) {
            hipLaunchKernelGGL(tr_scatter_add_kernel, grid1((size_t)N * H, 256), dim3(256), 0, st, (const float*)dhin, P.hin_w, 0, H,
                               (const int*)nullptr, dh0, H, 0, (long long)N);
            hipLaunchKernelGGL(tr_slice_kernel, grid1((size_t)N * H, 256), dim3(256), 0, st, (const float*)dhin, P.hin_w, H, H, dh,
                               (long long)N);
        } else {
            HIP_TRY(hipMemcpyAsync(dh, dhin, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(hipGetLastError());
    }
    // the initial latents feed step 1 directly and, when reattached, every step
    hipLaunchKernelGGL(bwd_add_kernel, grid1((size_t)N * H, 256), dim3(256), 0, st, dh, (const float*)dh0, (long long)N * H);
    hipLaunchKernelGGL(bwd_add_kernel, grid1((size_t)E * EF, 256), dim3(256), 0, st, de, (const float*)de0, (long long)E * EF);
    HIP_TRY(hipGetLastError());
    // encoders (no gradient flows into x / edge_attr)
    if (d->enc_edge.n_layers > 0) {
        float* g = gE0;
        HIP_TRY(hipMemcpyAsync(g, de, (size_t)E * EF * 4, hipMemcpyDeviceToDevice, st));
        s = tr_mlp_backward(c, P.enc_edge, edge_attr, g, gE0, gE1, false, nullptr);
        if (s != GNNCCA_OK) return s;
    }
    if (d->enc_node.n_layers > 0) {
        float* g = gN0;
        HIP_TRY(hipMemcpyAsync(g, dh, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
        s = tr_mlp_backward(c, P.enc_node, x, g, gN0, gN1, false, nullptr);
        if (s != GNNCCA_OK) return s;
    }
    return GNNCCA_OK;
}


// ============================================================================================================
// Stand-alone calls of the sub-modules (models/mlp.py:26-28, models/mpn.py:32-54,59-101,128-142 called on their own, which the
// reference allows): eval semantics (BatchNorm from the running statistics, Dropout = identity), the same one-launch-per-op
// kernels as the engine above.  MOTMPNet.forward never comes through here; gnn-cca_amd/mpn.py: _standalone_*.
// ============================================================================================================
__global__ void tr_bn_eval_stat_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ stat, int O) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < O) {
        stat[c] = rm[c];
        stat[O + c] = (float)(1.0 / sqrt((double)rv[c] + (double)kBnEps));
    }
}

// out[r][:] = cat(a[ia[r]], b[ib[r]], c[ic[r]]) with int64 row ids (null: the row itself); dense inputs (ld = width)
__global__ __launch_bounds__(256) void tr_cat64_kernel(const float* __restrict__ a, const long long* __restrict__ ia, int wa, long long na,
                                                       const float* __restrict__ b, const long long* __restrict__ ib, int wb, long long nb,
                                                       const float* __restrict__ cc, const long long* __restrict__ ic, int wc, long long nc,
                                                       float* __restrict__ out, long long M) {
    const int W = wa + wb + wc;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * W) return;
    const long long r = t / W;
    int col = (int)(t - r * W);
    const float* src = a;
    const long long* idx = ia;
    int w = wa;
    long long rows = na;
    if (col >= wa) {
        col -= wa, src = b, idx = ib, w = wb, rows = nb;
        if (col >= wb) col -= wb, src = cc, idx = ic, w = wc, rows = nc;
    }
    long long rr = idx ? idx[r] : r;
    rr = rr < 0 ? 0 : (rr >= rows ? rows - 1 : rr);   // out-of-range ids: stay in bounds (the reference raises an IndexError)
    out[t] = src[(size_t)rr * w + col];
}

static size_t mlp_eval_ws(const gnncca_mlp* m, int64_t rows, size_t* wt_off, size_t* bp_off, size_t* stat_off, size_t* buf_off) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    int maxo = 1;
    for (int l = 0; l < m->n_layers; ++l) {
        const int OP = (m->layers[l].out_dim + 7) / 8 * 8;
        const size_t w = take((size_t)m->layers[l].in_dim * OP * 4), b = take((size_t)OP * 4), st = take(2 * (size_t)m->layers[l].out_dim * 4);
        if (wt_off) wt_off[l] = w, bp_off[l] = b, stat_off[l] = st;
        maxo = std::max(maxo, (int)m->layers[l].out_dim);
    }
    for (int i = 0; i < 2; ++i) {
        const size_t o = take((size_t)std::max<int64_t>(rows, 0) * maxo * 4);
        if (buf_off) buf_off[i] = o;
    }
    return off;
}

static bool mlp_shape_ok(const gnncca_mlp* m) {
    if (!m || m->n_layers < 0 || m->n_layers > GNNCCA_MAX_LAYERS) return false;
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->layers[l].in_dim <= 0 || m->layers[l].out_dim <= 0) return false;
        if (l > 0 && m->layers[l].in_dim != m->layers[l - 1].out_dim) return false;
    }
    return true;
}

static int mlp_eval_impl(const gnncca_mlp* m, const float* const* params, int n_params, const float* in, int64_t rows, float* out,
                         void* ws, size_t ws_bytes, hipStream_t st) {
    if (!mlp_shape_ok(m) || !params || !in || !out || rows < 0 || rows > 0x7fffffffLL) return GNNCCA_ERR_INVALID_ARG;
    if (n_params != mlp_param_count(*m)) return GNNCCA_ERR_INVALID_ARG;
    size_t wt[GNNCCA_MAX_LAYERS], bp[GNNCCA_MAX_LAYERS], stat[GNNCCA_MAX_LAYERS], buf[2];
    if (ws_bytes < mlp_eval_ws(m, rows, wt, bp, stat, buf) || (!ws && m->n_layers > 0)) return GNNCCA_ERR_WORKSPACE;
    char* base = static_cast<char*>(ws);
    const long long M = rows;
    if (m->n_layers == 0 || M == 0) return GNNCCA_OK;
    const GenSeg none = {nullptr, nullptr, 0, 0};
    const float* cur = in;
    int pi = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        const gnncca_layer& L = m->layers[l];
        const int O = L.out_dim, K = L.in_dim, OP = (O + 7) / 8 * 8;
        float* Wt = reinterpret_cast<float*>(base + wt[l]);
        float* bpad = reinterpret_cast<float*>(base + bp[l]);
        hipLaunchKernelGGL(tr_transpose_pad_kernel, grid1((size_t)K * OP, 256), dim3(256), 0, st, params[pi], params[pi + 1], Wt, bpad, O, K, OP);
        const bool last = l == m->n_layers - 1;
        float* dst = last ? out : reinterpret_cast<float*>(base + buf[l & 1]);
        for (int o0 = 0; o0 < O; o0 += 2048) {
            const int og = std::min(2048, O - o0);
            const int rows_per_block = 256 / ((og + 7) / 8);
            hipLaunchKernelGGL(gen_dense_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st,
                               GenSeg{cur, nullptr, K, K}, none, none, (const float*)Wt + o0, (const float*)bpad + o0, dst + o0, M, K, og, OP, O,
                               (L.has_bn ? 0 : L.relu));
        }
        if (L.has_bn) {
            float* sp = reinterpret_cast<float*>(base + stat[l]);
            hipLaunchKernelGGL(tr_bn_eval_stat_kernel, dim3((O + 255) / 256), dim3(256), 0, st, params[pi + 4], params[pi + 5], sp, O);
            hipLaunchKernelGGL(tr_bn_apply_kernel, grid1((size_t)M * O, 256), dim3(256), 0, st, (const float*)dst, (const float*)sp,
                               params[pi + 2], params[pi + 3], dst, M, O, (int)L.relu);
        }
        HIP_TRY(hipGetLastError());
        cur = dst;
        pi += 2 + (L.has_bn ? 4 : 0);
    }
    return GNNCCA_OK;
}

struct AggWs { size_t flags, blockflags, seg_ptr, col32, perm, cursor, total; };
static AggWs agg_ws(int64_t n, int64_t e) {
    AggWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t N = (size_t)std::max<int64_t>(n, 0), E = (size_t)std::max<int64_t>(e, 0);
    w.flags = take(256), w.blockflags = take((E / 256 + 2) * 4), w.seg_ptr = take((N + 1) * 4), w.col32 = take(E * 4);
    w.perm = take(E * 4), w.cursor = take((N + 1) * 4), w.total = off;
    return w;
}

static int aggregate_impl(const float* msg, const int64_t* edge_index, int64_t n_nodes, int64_t n_edges, int H, int agg, float* out,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    if (!out || n_nodes < 0 || n_edges < 0 || n_nodes > 0x7fffffffLL || n_edges > 0x7fffffffLL || H <= 0) return GNNCCA_ERR_INVALID_ARG;
    if (agg != GNNCCA_AGG_SUM && agg != GNNCCA_AGG_MEAN && agg != GNNCCA_AGG_MAX) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges > 0 && (!msg || !edge_index)) return GNNCCA_ERR_INVALID_ARG;
    const AggWs w = agg_ws(n_nodes, n_edges);
    if (!ws || ws_bytes < w.total) return GNNCCA_ERR_WORKSPACE;
    if (n_nodes == 0) return GNNCCA_OK;
    char* base = static_cast<char*>(ws);
    const int N = (int)n_nodes, E = (int)n_edges;
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    unsigned* flags = reinterpret_cast<unsigned*>(base + w.flags);
    unsigned* blockflags = reinterpret_cast<unsigned*>(base + w.blockflags);
    int* seg_ptr = reinterpret_cast<int*>(base + w.seg_ptr);
    int* col32 = reinterpret_cast<int*>(base + w.col32);
    int* perm = reinterpret_cast<int*>(base + w.perm);
    int* cursor = reinterpret_cast<int*>(base + w.cursor);
    if (E > 0) {
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.ei = ei, ep.seg_ptr = seg_ptr, ep.col32 = col32, ep.blockflags = blockflags, ep.E = E, ep.N = N;
        hipLaunchKernelGGL(plan_only_kernel, dim3(plan_num_blocks(E)), dim3(256), 0, st, ep);
    }
    hipLaunchKernelGGL(gen_plan_finish_kernel, dim3(1), dim3(256), 0, st, ei, E, N, seg_ptr, col32, perm, cursor, flags,
                       (const unsigned*)blockflags);
    hipLaunchKernelGGL(gen_aggregate_kernel, dim3((unsigned)N), dim3(256), 0, st, msg, (const int*)seg_ptr, (const int*)perm,
                       (const unsigned*)flags, out, N, H, agg);
    HIP_TRY(hipGetLastError());
    return GNNCCA_OK;
}

}  // namespace gnncca
