// mpn_kernels.hip -- gfx950 (MI355X / CDNA4) kernels and the C-ABI forward of the GNN-CCA message-passing path.
//
// Algebra (derived from models/mpn.py:48,68-69,97-99): with the edge-MLP weight split by the
// cat order [x[row] | x[col] | e] and the node-MLP weight by [x[row] | e'],
//     P_src = h W_src^T + b_e,  P_dst = h W_dst^T,  Q = h W_nx^T + b_n            (per node, tiny)
//     e'[k] = ReLU(P_src[row k] + P_dst[col k] + W_ee e[k])                         (per edge, VALU)
//     m[k]  = ReLU(Q[row k] + W_ne e'[k])                                           (per edge, MFMA 32x32x2 f32)
//     h'[i] = agg_{k : row k = i} m[k]                                              (in-register, per segment)
// so no [E,70] / [E,38] concatenation is ever materialised.  The aggregation index is `row` (the SOURCE node),
// exactly as the reference does it (mpn.py:99).
//
// Data layout in HBM (all fp32):
//   edge state   e      : 6 feature planes [6][E_pad]  in ROW-SORTED edge order  -> coalesced 256-B wave loads
//   gather table Pd     : [N][8]   (P_dst, 32-B rows)                             -> L1/L2-resident random reads
//   segment table PsQ   : [N][40]  (P_src | pad | Q)                              -> wave-uniform reads
//   topology     seg_ptr: [N+1] int32 CSR offsets by source node;  col32 [E] int32 (sorted order)
// One wave owns (a share of) one source node's contiguous edge segment, so the per-destination reduction needs
// no atomics and is bitwise reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "internal.h"

namespace gnncca {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

thread_local int g_last_hip_error = 0;

// Diagnostic build only (-DGNNCCA_STAMPS, tools/stamps.py): s_memtime stamps of every wave at named points, written to
// a buffer of their own that no kernel reads.  The product build compiles these to nothing.
#ifdef GNNCCA_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define GNNCCA_STAMP(kslot, id)                                                                              \
    do {                                                                                                     \
        if (g_stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) {                                      \
            g_stamps[((((size_t)(kslot)) * 4096 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + (id)] =       \
                __builtin_amdgcn_s_memtime();                                                                \
        }                                                                                                    \
    } while (0)
#else
#define GNNCCA_STAMP(kslot, id) \
    do {                        \
    } while (0)
#endif

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_last_hip_error = (int)_e;                \
            return GNNCCA_ERR_HIP;                     \
        }                                              \
    } while (0)

// Per-kernel timing for bench.py / rocprof cross-checks: one hipEvent after every launch (diagnostic entry point only).
struct Profiler {
    gnncca_profile* out;
    hipEvent_t ev[GNNCCA_PROFILE_MAX + 1];
    int n;
};

static int prof_begin(Profiler* p, hipStream_t st) {
    if (!p) return GNNCCA_OK;
    p->n = 0;
    for (int i = 0; i <= GNNCCA_PROFILE_MAX; ++i) HIP_TRY(hipEventCreate(&p->ev[i]));
    HIP_TRY(hipEventRecord(p->ev[0], st));
    return GNNCCA_OK;
}

static int prof_mark(Profiler* p, int kind, hipStream_t st) {
    if (!p || p->n >= GNNCCA_PROFILE_MAX) return GNNCCA_OK;
    p->out->kind[p->n] = kind;
    p->n++;
    HIP_TRY(hipEventRecord(p->ev[p->n], st));
    return GNNCCA_OK;
}

static int prof_end(Profiler* p, hipStream_t st) {
    if (!p) return GNNCCA_OK;
    HIP_TRY(hipStreamSynchronize(st));
    p->out->count = p->n;
    for (int i = 0; i < p->n; ++i) HIP_TRY(hipEventElapsedTime(&p->out->ms[i], p->ev[i], p->ev[i + 1]));
    for (int i = 0; i <= GNNCCA_PROFILE_MAX; ++i) HIP_TRY(hipEventDestroy(p->ev[i]));
    return GNNCCA_OK;
}

#define PROF_MARK(kind)                                  \
    do {                                                 \
        int _s = prof_mark(prof, (kind), st);            \
        if (_s != GNNCCA_OK) return _s;                  \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// Graph plan.  Parallel, optimistic part: validates indices, narrows `col` to int32 and builds the CSR offsets
// assuming `row` is non-decreasing -- true for every graph the reference builds (inference.py:209-216;
// Batch.from_data_list keeps the order).  A violation raises GNNCCA_GRAPH_UNSORTED; one extra workgroup of the
// next launch on the stream (enc_tail_kernel) then repairs the plan alone with a STABLE counting sort by `row`
// (plan_sort_fallback), so that every segment keeps the caller's edge order -- the order torch's CPU index_add_
// (and with it the reference on CPU) sums in.  Only correctness matters on that branch: the reference never
// produces such graphs.  The kernel boundary orders it after the plan's stores; no in-launch hand-off.
// ------------------------------------------------------------------------------------------------------------
__device__ void plan_sort_fallback(const long long* __restrict__ ei, int E, int N, int* seg_ptr, int* col32, int* perm,
                                   int* cursor, int* s_rows, int* s_scan, int* s_carry) {
    constexpr int B = 256;
    const int tid = threadIdx.x;
    for (int n = tid; n <= N; n += B) __hip_atomic_store(&cursor[n], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) *s_carry = 0;
    __syncthreads();
    for (int k = tid; k < E; k += B) atomicAdd(&cursor[(int)ei[k]], 1);
    __syncthreads();
    // exclusive scan of the histogram -> seg_ptr; cursor[n] := seg_ptr[n]
    for (int n0 = 0; n0 <= N; n0 += B) {
        const int n = n0 + tid;
        const int v = (n < N) ? __hip_atomic_load(&cursor[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int d = 1; d < B; d <<= 1) {
            const int add = (tid >= d) ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        const int excl = *s_carry + s_scan[tid] - v;
        if (n <= N) {
            seg_ptr[n] = excl;
            __hip_atomic_store(&cursor[n], excl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid == B - 1) *s_carry += s_scan[B - 1];
        __syncthreads();
    }
    // stable placement, B edges at a time in ascending edge id
    for (int k0 = 0; k0 < E; k0 += B) {
        const int k = k0 + tid;
        const int r = (k < E) ? (int)ei[k] : -1;
        s_rows[tid] = r;
        __syncthreads();
        if (k < E) {
            int rank = 0;
            for (int u = 0; u < tid; ++u) rank += (s_rows[u] == r);
            const int pos = __hip_atomic_load(&cursor[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + rank;
            perm[pos] = k;
            col32[pos] = (int)ei[(size_t)E + k];
        }
        __syncthreads();
        if (k < E) atomicAdd(&cursor[r], 1);
        __syncthreads();
    }
}

// Thread-per-edge part of the plan.  Every workgroup reports its findings in its OWN word (`blockflags[b]`, always
// written, so there is no state to clear between forwards); the tail launch ORs them into flags[0].
__device__ __forceinline__ void plan_block(int pb, const long long* __restrict__ ei, int E, int N,
                                           int* __restrict__ seg_ptr, int* __restrict__ col32,
                                           unsigned* __restrict__ blockflags, unsigned* s_fl) {
    if (threadIdx.x == 0) *s_fl = 0u;
    __syncthreads();
    const int k = pb * 256 + threadIdx.x;
    if (k < E) {
        const long long r = ei[k], c = ei[(size_t)E + k];
        if (r < 0 || r >= N || c < 0 || c >= N) {
            atomicOr(s_fl, GNNCCA_GRAPH_BAD_INDEX);
        } else {
            col32[k] = (int)c;
            long long rp = -1;
            bool prev_ok = true;
            if (k > 0) {
                rp = ei[k - 1];
                prev_ok = rp >= 0 && rp < N;  // otherwise its owner raises the flag
            }
            if (prev_ok) {
                if (r < rp) {
                    atomicOr(s_fl, GNNCCA_GRAPH_UNSORTED);
                } else {
                    for (long long n = rp + 1; n <= r; ++n) seg_ptr[n] = k;
                }
                if (k == E - 1)
                    for (long long n = r + 1; n <= N; ++n) seg_ptr[n] = E;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) blockflags[pb] = *s_fl;
}

// ------------------------------------------------------------------------------------------------------------
// Node encoder GEMM: part[ks][M][O] = in[M][kslice ks] . W[O][kslice ks]^T with v_mfma_f32_32x32x2_f32 (exact
// fp32 FMA chain).  One wave = 32 rows x 32 output columns; a workgroup = 4 waves = 128 columns.
// k-permutation: within a 64-deep chunk, lane (r, h) feeds k = kc + 32h + s at MFMA step s for BOTH operands, so
// every lane reads 128 contiguous bytes of its own row (8 x float4) and no LDS transpose is needed.
// Split-K fills the chip when M is small (M = 256 nodes -> 8 row tiles x 32 slices).
// Replaces the first nn.Linear of encoder.node_mlp (models/mpn.py:131 <- models/mlp.py:13).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gemm_tile(int rt, int ks, int cg, const float* __restrict__ in, const float* __restrict__ W,
                                          float* __restrict__ part, int M, int K, int O, int kslice, int vec_ok) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = rt * 32;
    const int col0 = (cg * 4 + wave) * 32;
    if (col0 >= O) return;
    const int arow = min(row0 + r, M - 1);
    const int wrow = min(col0 + r, O - 1);
    const float* __restrict__ ap = in + (size_t)arow * K;
    const float* __restrict__ wp = W + (size_t)wrow * K;
    const int kbeg = ks * kslice;
    const int kend = min(kbeg + kslice, K);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int kc = kbeg; kc < kend; kc += 64) {
        float a[32], b[32];
        const int k0 = kc + 32 * h;
        if (vec_ok && kc + 64 <= kend) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k0 + 4 * j);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wp + k0 + 4 * j);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[4 * j + q] = av[q];
                    b[4 * j + q] = bv[q];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int k = k0 + s;
                a[s] = (k < kend) ? ap[k] : 0.f;
                b[s] = (k < kend) ? wp[k] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    const int col = col0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M && col < O) part[((size_t)ks * M + row) * O + col] = acc[i];
    }
}

// Second half of the plan, run by ONE 256-thread workgroup of the launch that follows the plan blocks on the stream:
// OR of the per-block findings -> flags[0]; stable counting sort if the rows were not sorted.  `smem` >= 3 KB.
__device__ void plan_finish(const long long* __restrict__ ei, int E, int N, int* seg_ptr, int* col32, int* perm,
                            int* cursor, unsigned* flags, const unsigned* __restrict__ blockflags, unsigned* smem) {
    const int tid = threadIdx.x;
    unsigned fl = 0u;
    const int nb = (E + 255) / 256;
    for (int i = tid; i < nb; i += 256) fl |= blockflags[i];
    smem[tid] = fl;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) smem[tid] |= smem[tid + d];
        __syncthreads();
    }
    fl = smem[0];
    __syncthreads();
    if (tid == 0) flags[0] = fl;
    if ((fl & GNNCCA_GRAPH_UNSORTED) && !(fl & GNNCCA_GRAPH_BAD_INDEX)) {
        int* si = reinterpret_cast<int*>(smem);
        plan_sort_fallback(ei, E, N, seg_ptr, col32, perm, cursor, si, si + 256, si + 512);
    }
}

// One launch, two roles: workgroups [0, gemm_blocks) run encoder GEMM tiles, the rest run the graph plan -- the
// two are independent, so the plan's HBM pass over edge_index hides under the GEMM instead of costing a launch.
struct EncPlanParams {
    const float* in;
    const float* W;
    float* part;
    const long long* ei;
    int* seg_ptr;
    int* col32;
    unsigned* blockflags;
    int M, K, O, kslice, vec_ok, nrt, nks, gemm_blocks, E, N;
};

__global__ __launch_bounds__(256) void enc_gemm_plan_kernel(const EncPlanParams p) {
    __shared__ unsigned s_fl;
    GNNCCA_STAMP(1, 0);
    const int b = blockIdx.x;
    if (b < p.gemm_blocks) {
        const int rt = b % p.nrt, t = b / p.nrt;
        gemm_tile(rt, t % p.nks, t / p.nks, p.in, p.W, p.part, p.M, p.K, p.O, p.kslice, p.vec_ok);
    } else {
        plan_block(b - p.gemm_blocks, p.ei, p.E, p.N, p.seg_ptr, p.col32, p.blockflags, &s_fl);
    }
    GNNCCA_STAMP(1, 1);
}

// ------------------------------------------------------------------------------------------------------------
// Large-N encoder GEMM on the bf16 MFMA pipe with fp32-level accuracy ("split-bf16"):  x = x0 + x1 + x2 and
// w = w0 + w1 + w2 with bf16 pieces (3 x 8 = 24 mantissa bits, the pieces of w prepared at pack time, those of x
// on the fly while staging), and  x.w ~= x2w0 + x1w1 + x0w2 + x1w0 + x0w1 + x0w0  -- the dropped terms are
// <= 2^-24 relative.  Every product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32, so the
// result differs from an fp32 FMA chain only by rounding of the same order as fp32 itself, while the six
// v_mfma_f32_32x32x16_bf16 cost 6/16 of the v_mfma_f32_32x32x2_f32 time: the GEMM becomes HBM-bound on the
// x read (8 KB per node) instead of MFMA-bound.
// Workgroup = 64 rows x 128 columns, 4 waves as 2 x 2 (32 rows x 64 columns each), K in chunks of 32 through LDS
// (rows padded to 80 B: conflict-free ds_read_b128); the next chunk's global loads are in flight during the MFMAs.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int WM>  // rows per wave: 32 (workgroup 64 x 128) or 64 (workgroup 128 x 128)
__global__ __launch_bounds__(256) void enc_gemm_split_kernel(const float* __restrict__ x, const unsigned short* __restrict__ w3,
                                                             float* __restrict__ out, int M, int K, int O, int kslice) {
    constexpr int BM = 2 * WM, BN = 128, BK = 32, LDK = 40;  // LDK: padded row length in bf16 elements (80 B)
    constexpr int RT = WM / 32;                              // 32-row MFMA tiles per wave
    constexpr int XU = BM * BK / 4 / 256;                    // float4 loads of x per thread per chunk
    __shared__ __attribute__((aligned(16))) __bf16 xs[3][BM][LDK];
    __shared__ __attribute__((aligned(16))) __bf16 wsm[3][BN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.x * BM;
    const int kbeg = blockIdx.y * kslice;
    const size_t plane = (size_t)O * K;
    int xr[XU], xc[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int idx = tid + 256 * u;
        xr[u] = idx >> 3;
        xc[u] = (idx & 7) * 4;
    }
    int wp[6], wcol[6], wk[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int idx = tid + 256 * u;
        wp[u] = idx >> 9;
        wcol[u] = (idx & 511) >> 2;
        wk[u] = (idx & 3) * 8;
    }
    f32x4 xreg[XU];
    bf16x8 wreg[6];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int r = min(row0 + xr[u], M - 1);
            xreg[u] = *reinterpret_cast<const f32x4*>(x + (size_t)r * K + kbeg + kt * BK + xc[u]);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u)
            wreg[u] = *reinterpret_cast<const bf16x8*>(w3 + wp[u] * plane + (size_t)wcol[u] * K + kbeg + kt * BK + wk[u]);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            bf16x4 p0, p1, p2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = xreg[u][q];
                const __bf16 h0 = (__bf16)v;
                const float r1 = v - (float)h0;
                const __bf16 h1 = (__bf16)r1;
                const float r2 = r1 - (float)h1;
                p0[q] = h0;
                p1[q] = h1;
                p2[q] = (__bf16)r2;
            }
            *reinterpret_cast<bf16x4*>(&xs[0][xr[u]][xc[u]]) = p0;
            *reinterpret_cast<bf16x4*>(&xs[1][xr[u]][xc[u]]) = p1;
            *reinterpret_cast<bf16x4*>(&xs[2][xr[u]][xc[u]]) = p2;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<bf16x8*>(&wsm[wp[u]][wcol[u]][wk[u]]) = wreg[u];
    };
    f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
    const int nk = min(kslice, K - kbeg) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    const int k8 = 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[RT][3], b[2][3];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[r][p] = *reinterpret_cast<const bf16x8*>(&xs[p][wr * WM + r * 32 + (lane & 31)][ks * 16 + k8]);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[c][p] = *reinterpret_cast<const bf16x8*>(&wsm[p][wc * 64 + c * 32 + (lane & 31)][ks * 16 + k8]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // smallest terms first
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][2], b[c][0], acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][1], b[c][1], acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][0], b[c][2], acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][1], b[c][0], acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][0], b[c][1], acc[r][c], 0, 0, 0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][0], b[c][0], acc[r][c], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) store_tile();
        __syncthreads();
    }
    float* __restrict__ dst = out + (size_t)blockIdx.y * M * O;  // split-K partial slab
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = wc * 64 + c * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + wr * WM + r * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (row < M) dst[(size_t)row * O + col] = acc[r][c][i];
            }
        }
}

// act[M][O] = [ReLU](bias + sum_ks part[ks][M][O]) -- only for encoders deeper than two layers.
__global__ __launch_bounds__(256) void reduce_bias_act_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              float* __restrict__ act, int M, int O, int ks, int relu) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * O) return;
    float v = bias[idx % O];
    for (int s = 0; s < ks; ++s) v += part[(size_t)s * M * O + idx];
    act[idx] = relu ? fmaxf(v, 0.f) : v;
}

// ------------------------------------------------------------------------------------------------------------
// Per-node projection for the NEXT message-passing step.  Called by one whole wave that holds the node's latent
// h[c] in lane c (c < 32, mirrored in lanes 32..63).  Lane o < 48 produces projection slot o:
//   [0,6) P_dst   [8,14) P_src + b_e   [16,48) Q + b_n          (weights transposed in LDS: [c][48])
// With reattach_initial_nodes the input is cat((initial, latent)) -- initial first (models/mpn.py:285).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void project_node(float h_latent, float h_init, bool reatt_n, const float* s_projT,
                                             const float* projb, float* __restrict__ pd_row,
                                             float* __restrict__ psq_row, int lane) {
    const int o = min(lane, kProjOut - 1);
    float acc = projb[o];
    const float* w = s_projT + o;
    if (reatt_n) {
#pragma unroll
        for (int c = 0; c < kH; ++c)
            acc = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_init), c)), acc);
        w += kH * kProjOut;
    }
#pragma unroll
    for (int c = 0; c < kH; ++c)
        acc = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_latent), c)), acc);
    if (lane < kPdStride)
        pd_row[lane] = acc;
    else if (lane < kProjOut)
        psq_row[lane - kPdStride] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// Encoder tail: finishes the previous GEMM layer (sum of split-K partials in fixed order + bias + ReLU), applies
// the last encoder layer F -> 32 (+ReLU), stores h0 and emits the step-1 projections.  One wave per node.
// Replaces the rest of encoder.node_mlp (models/mpn.py:131) and the x[row]/x[col] gathers of step 1.
// ------------------------------------------------------------------------------------------------------------
struct TailParams {
    const float* blob;
    const float* part;
    float* h0;
    float* trace_h;
    float* pd_out;
    float* psq_out;
    int off_prev_b, off_lastWT, off_last_b, off_projwT, off_projb;
    int ks, F, N, has_last, relu_prev, reatt_n, hin, vec_reduce;
    // graph-plan repair (runs in the extra last workgroup only when the graph was flagged unsorted)
    const long long* ei;
    int* seg_ptr;
    int* col32;
    int* perm;
    int* cursor;
    unsigned* flags;
    const unsigned* blockflags;
    int E;
};

__global__ __launch_bounds__(256) void enc_tail_kernel(const TailParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                    // [hin][48]
    float* s_last = s_proj + p.hin * kProjOut;               // [F][32]   (has_last)
    float* s_row = s_last + (p.has_last ? p.F * kH : 0);     // [4][F]
    float* s_red = s_row + 4 * p.F;                          // [4][64 floats x 4]  (vec_reduce)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ blob = p.blob;
    if (blockIdx.x == gridDim.x - 1) {  // the plan workgroup: fold the per-block findings, repair if needed
        plan_finish(p.ei, p.E, p.N, p.seg_ptr, p.col32, p.perm, p.cursor, p.flags, p.blockflags,
                    reinterpret_cast<unsigned*>(smem));
        return;
    }
    GNNCCA_STAMP(0, 0);
    const int nblk = gridDim.x - 1;
    const int o = lane & 31, half = lane >> 5;
    float* rowbuf = s_row + wave * p.F;
    float* redbuf = s_red + wave * 256;
    // split-K partial sum of one node row: F/4 float4 chunks per row; 64/(F/4) lane groups walk the partials in an
    // interleaved, fixed order with all loads independent (one round trip instead of ks dependent ones)
    const int nchunk = p.vec_reduce ? (p.F >> 2) : 64, groups = 64 / nchunk;
    const int rg = lane / nchunk, rc = lane - rg * nchunk;
    auto partial_sum = [&](int node) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (p.vec_reduce && node < p.N && rg < groups) {
            const float* __restrict__ src = p.part + (size_t)node * p.F + 4 * rc;
            const size_t sstride = (size_t)p.N * p.F;
#pragma unroll 8
            for (int s = rg; s < p.ks; s += groups) a += *reinterpret_cast<const f32x4*>(src + s * sstride);
        }
        return a;
    };
    // Issue order = completion order (vmcnt): weights first (consumed first, by the LDS stage), then biases, then
    // the first node's partials, so that one round trip covers all three.
    const int n4p = p.hin * kProjOut / 4, n4l = p.has_last ? p.F * kH / 4 : 0;
    const f32x4* __restrict__ gp4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
    const f32x4* __restrict__ gl4 = reinterpret_cast<const f32x4*>(blob + p.off_lastWT);
    f32x4 rp[3], rl[4];
#pragma unroll
    for (int u = 0; u < 3; ++u) rp[u] = gp4[min(u * 256 + tid, n4p - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) rl[u] = gl4[min(u * 256 + tid, max(n4l - 1, 0))];
    const float last_b = p.has_last ? blob[p.off_last_b + o] : 0.f;
    const float prev_b0 = blob[p.off_prev_b + min(lane, p.F - 1)];
    const float prev_b1 = blob[p.off_prev_b + min(lane + 64, p.F - 1)];
    f32x4 pre = partial_sum(blockIdx.x * 4 + wave);
    GNNCCA_STAMP(0, 1);
    {
        f32x4* lp4 = reinterpret_cast<f32x4*>(s_proj);
        f32x4* ll4 = reinterpret_cast<f32x4*>(s_last);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u * 256 + tid < n4p) lp4[u * 256 + tid] = rp[u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u * 256 + tid < n4l) ll4[u * 256 + tid] = rl[u];
        for (int i = 1024 + tid; i < n4l; i += 256) ll4[i] = gl4[i];  // F > 128: the rest, plainly
    }
    GNNCCA_STAMP(0, 2);
    __syncthreads();
    GNNCCA_STAMP(0, 3);
    for (int grp = blockIdx.x; grp * 4 < p.N; grp += nblk) {
        const int node = grp * 4 + wave;
        const bool active = node < p.N;
        if (active) {
            if (p.vec_reduce) {
                if (rg < groups) *reinterpret_cast<f32x4*>(redbuf + rg * p.F + 4 * rc) = pre;
                __builtin_amdgcn_wave_barrier();
                for (int f = lane; f < p.F; f += 64) {
                    float v = f < 64 ? prev_b0 : (f < 128 ? prev_b1 : blob[p.off_prev_b + f]);
                    for (int gg = 0; gg < groups; ++gg) v += redbuf[gg * p.F + f];
                    rowbuf[f] = p.relu_prev ? fmaxf(v, 0.f) : v;
                }
            } else {
                for (int f = lane; f < p.F; f += 64) {
                    float v = blob[p.off_prev_b + f];
                    for (int s = 0; s < p.ks; ++s) v += p.part[((size_t)s * p.N + node) * p.F + f];
                    rowbuf[f] = p.relu_prev ? fmaxf(v, 0.f) : v;
                }
            }
        }
        GNNCCA_STAMP(0, 4);
        pre = partial_sum((grp + nblk) * 4 + wave);  // next node of this wave, in flight during the layer below
        __syncthreads();
        GNNCCA_STAMP(0, 5);
        float hv = 0.f;
        if (active) {
            if (p.has_last) {
                // four independent accumulators and an 8-deep unroll keep 16 LDS reads in flight per lane
                float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
                int f = half;
                for (; f + 14 < p.F; f += 16) {
#pragma unroll
                    for (int u = 0; u < 8; u += 4) {
                        acc0 = fmaf(rowbuf[f + 2 * u + 0], s_last[(f + 2 * u + 0) * kH + o], acc0);
                        acc1 = fmaf(rowbuf[f + 2 * u + 2], s_last[(f + 2 * u + 2) * kH + o], acc1);
                        acc2 = fmaf(rowbuf[f + 2 * u + 4], s_last[(f + 2 * u + 4) * kH + o], acc2);
                        acc3 = fmaf(rowbuf[f + 2 * u + 6], s_last[(f + 2 * u + 6) * kH + o], acc3);
                    }
                }
                for (; f < p.F; f += 2) acc0 = fmaf(rowbuf[f], s_last[f * kH + o], acc0);
                float acc = (acc0 + acc1) + (acc2 + acc3);
                acc += __shfl_xor(acc, 32);
                hv = fmaxf(acc + last_b, 0.f);
            } else {
                hv = rowbuf[o];
            }
            GNNCCA_STAMP(0, 6);
            if (lane < kH) {
                p.h0[(size_t)node * kH + lane] = hv;
                if (p.trace_h) p.trace_h[(size_t)node * kH + lane] = hv;
            }
            project_node(hv, hv, p.reatt_n != 0, s_proj, blob + p.off_projb, p.pd_out + (size_t)node * kPdStride,
                         p.psq_out + (size_t)node * kPsQStride, lane);
            GNNCCA_STAMP(0, 7);
        }
        __syncthreads();
    }
    GNNCCA_STAMP(0, 8);
}

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
    const unsigned* flags;
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
    long long e_stride;
    int off_wee, off_wneb, off_projwT, off_projb, off_encw, off_encb, off_cw1, off_cb1, off_cw2, off_cb2;
    int off_fast;
    int cls_layers, cls_hidden;  // cls_layers == 0: this step does not classify
    int N, E, edge_in, attr_vec, first, update, agg, reatt_n, wps, store_e, hin, pd_lds, stamp_slot;
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
    {
        const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = p.update ? psq[f] : 0.f;
        if (MSG) {
            cinit = psq[8 + ch];
#pragma unroll
            for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
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
                    logit = fmaf(blob[p.off_cw2 + q], fmaxf(z, 0.f), logit);
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
            // accumulator register i of lane (ch, half) is edge (i&3) + 8*(i>>2) + 4*half of the tile
            if (base + 64 <= seg_t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float m0 = fmaxf(d0[i], 0.f), m1 = fmaxf(d1[i], 0.f);
                    acc[i] = agg_max ? fmaxf(acc[i], fmaxf(m0, m1)) : acc[i] + (m0 + m1);
                }
            } else {
                const int rem = seg_t - base - 4 * half;
                const float ident = agg_max ? -INFINITY : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int eo = (i & 3) + 8 * (i >> 2);
                    const float m0 = (eo < rem) ? fmaxf(d0[i], 0.f) : ident;
                    const float m1 = (eo + 32 < rem) ? fmaxf(d1[i], 0.f) : ident;
                    acc[i] = agg_max ? fmaxf(acc[i], fmaxf(m0, m1)) : acc[i] + (m0 + m1);
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
// Specialised step kernel for the shipped GRAPH_NET_PARAMS shape (edge_in 4, no reattach, classifier 6->4->1 or
// off, no debug taps): the same algorithm as mpn_step_kernel with every per-config decision made at compile
// time, so the body is straight-line code whose loads issue back to back.  mpn_step_kernel stays as the
// general / traced variant.
// ------------------------------------------------------------------------------------------------------------
template <bool FIRST, bool CLS, bool MSG, bool PD_LDS>
__global__ __launch_bounds__(256) void mpn_step_fast_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_part = s_proj + (MSG ? kH * kProjOut : 0);     // [4][32]
    float* s_pd = s_part + 4 * kH;                          // [N][8]     (PD_LDS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    // Per-step scalars (152 floats) are read through the CONSTANT address space: wave-uniform addresses there
    // become s_load into SGPRs, which the VALU takes as operands directly -- no LDS, no VGPR copies.
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);

    GNNCCA_STAMP(p.stamp_slot, 0);
    // ---- prologue: every independent load is issued before the first wait ----------------------------------
    const unsigned gflags = p.flags[0];
    const int wps = p.wps;
    const int node = blockIdx.x * (4 / wps) + wave / wps;
    const int sub = wave % wps;
    const bool active = node < p.N;
    const int nclamp = active ? node : 0;
    int seg_s = p.seg_ptr[nclamp];
    int seg_t = p.seg_ptr[nclamp + 1];
    const int half = lane >> 5, ch = lane & 31;
    const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
    float psrc[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
    float cinit = 0.f;
    float bw[3] = {0.f, 0.f, 0.f};
    f32x4 stage_proj[2];
    f32x4 stage_pd[8];
    float projb_l = 0.f;
    if (MSG) {
        cinit = psq[8 + ch];
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
#pragma unroll
        for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        stage_proj[0] = g4[tid];                                   // 384 float4 in all
        stage_proj[1] = g4[min(tid + 256, kH * kProjOut / 4 - 1)];
    }
    const int pd_n4 = p.N * (kPdStride / 4);
    if (PD_LDS) {  // N <= 1024: at most 8 float4 per thread
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.pd_in);
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_pd[i] = g4[min(tid + i * 256, pd_n4 - 1)];
    }
    if (MSG) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        l4[tid] = stage_proj[0];
        if (tid + 256 < kH * kProjOut / 4) l4[tid + 256] = stage_proj[1];
    }
    if (PD_LDS) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_pd);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (tid + i * 256 < pd_n4) l4[tid + i * 256] = stage_pd[i];
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    GNNCCA_STAMP(p.stamp_slot, 1);
    if (MSG || PD_LDS) __syncthreads();
    GNNCCA_STAMP(p.stamp_slot, 2);
    if (!active) seg_s = seg_t = 0;

    f32x16 acc;  // 'sum' / 'mean' only: 'max' takes the general kernel
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int last = seg_t - 1;

    struct Chunk {
        float raw[kEF];
        float pd[kEF];
        int ko;
    };
    auto load_chunk = [&](int base, Chunk& c) {
        const int kk = min(base + lane, last);
        c.ko = unsorted ? p.perm[kk] : kk;
        const int j = p.col32[kk];
        if (FIRST) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.edge_attr + (size_t)c.ko * 4);
            c.raw[0] = a[0], c.raw[1] = a[1], c.raw[2] = a[2], c.raw[3] = a[3], c.raw[4] = 0.f, c.raw[5] = 0.f;
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) c.raw[f] = p.e[(size_t)f * p.e_stride + kk];
        }
        f32x4 a;
        f32x2 b2;
        if (PD_LDS) {
            a = *reinterpret_cast<const f32x4*>(s_pd + j * kPdStride);
            b2 = *reinterpret_cast<const f32x2*>(s_pd + j * kPdStride + 4);
        } else {
            const float* __restrict__ pdj = p.pd_in + (size_t)j * kPdStride;
            a = *reinterpret_cast<const f32x4*>(pdj);
            b2 = *reinterpret_cast<const f32x2*>(pdj + 4);
        }
        c.pd[0] = a[0], c.pd[1] = a[1], c.pd[2] = a[2], c.pd[3] = a[3], c.pd[4] = b2[0], c.pd[5] = b2[1];
    };
    auto compute_chunk = [&](int base, const Chunk& c) {
        const int k = base + lane;
        const bool valid = k < seg_t;
        float ein[kEF];
        if (FIRST) {
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                float s = cw[kFcEncB + f];
#pragma unroll
                for (int q = 0; q < 4; ++q) s = fmaf(cw[kFcEncW + f * 4 + q], c.raw[q], s);
                ein[f] = fmaxf(s, 0.f);
            }
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) ein[f] = c.raw[f];
        }
        float en[kEF];
#pragma unroll
        for (int f = 0; f < kEF; ++f) {
            float s = psrc[f] + c.pd[f];
#pragma unroll
            for (int g = 0; g < kEF; ++g) s = fmaf(cw[kFcWee + f * kEF + g], ein[g], s);
            en[f] = fmaxf(s, 0.f);
        }
        if (p.store_e && valid) {
#pragma unroll
            for (int f = 0; f < kEF; ++f) p.e[(size_t)f * p.e_stride + k] = en[f];
        }
        if (CLS) {
            float logit = cw[kFcCb2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z = cw[kFcCb1 + q];
#pragma unroll
                for (int f = 0; f < kEF; ++f) z = fmaf(cw[kFcCw1 + q * kEF + f], en[f], z);
                logit = fmaf(cw[kFcCw2 + q], fmaxf(z, 0.f), logit);
            }
            if (valid) p.logits[c.ko] = logit;
        }
        if (MSG) {
            f32x16 d0, d1;
#pragma unroll
            for (int i = 0; i < 16; ++i) d0[i] = d1[i] = cinit;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(en[2 * s]), __float_as_uint(en[2 * s + 1]),
                                                                false, false);
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[0]), bw[s], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[1]), bw[s], d1, 0, 0, 0);
            }
            if (base + 64 <= seg_t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += fmaxf(d0[i], 0.f) + fmaxf(d1[i], 0.f);
            } else {
                const int rem = seg_t - base - 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int eo = (i & 3) + 8 * (i >> 2);
                    const float m0 = (eo < rem) ? fmaxf(d0[i], 0.f) : 0.f;
                    const float m1 = (eo + 32 < rem) ? fmaxf(d1[i], 0.f) : 0.f;
                    acc[i] += m0 + m1;
                }
            }
        }
    };

    // Two chunks (128 edges, 3.6 KB of loads) are requested before the first one is consumed.
    const int stride = 64 * wps;
    for (int base = seg_s + 64 * sub; base < seg_t; base += 2 * stride) {
        Chunk c0, c1;
        const bool two = base + stride < seg_t;
        load_chunk(base, c0);
        if (two) load_chunk(base + stride, c1);
        GNNCCA_STAMP(p.stamp_slot, 3);
        compute_chunk(base, c0);
        if (two) compute_chunk(base + stride, c1);
        GNNCCA_STAMP(p.stamp_slot, 4);
    }
    GNNCCA_STAMP(p.stamp_slot, 5);

    if (MSG) {
        float v = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += acc[i];
        v += __shfl_xor(v, 32);
        if (wps > 1) {
            if (lane < kH) s_part[wave * kH + lane] = v;
            __syncthreads();
            if (sub == 0) {
                v = s_part[wave * kH + ch];
                for (int u = 1; u < wps; ++u) v += s_part[(wave + u) * kH + ch];
            }
        }
        GNNCCA_STAMP(p.stamp_slot, 6);
        if (active && sub == 0) {
            const int deg = seg_t - seg_s;
            if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
            if (deg == 0) v = 0.f;
            // projection epilogue (project_node with the bias read through the constant address space)
            const int o = min(lane, kProjOut - 1);
            float pr = projb_l;
            const float* w = s_proj + o;
#pragma unroll
            for (int c = 0; c < kH; ++c)
                pr = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)), pr);
            if (lane < kPdStride)
                p.pd_out[(size_t)node * kPdStride + lane] = pr;
            else if (lane < kProjOut)
                p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
        }
    }
    GNNCCA_STAMP(p.stamp_slot, 7);
}

// ------------------------------------------------------------------------------------------------------------
static inline dim3 grid1(size_t n, int b) { return dim3((unsigned)((n + b - 1) / b)); }

template <bool RE, bool MSG, bool MX>
static hipError_t launch_step_t(const StepParams& sp, hipStream_t st) {
    const int npg = 4 / sp.wps;
    const unsigned blocks = (unsigned)((sp.N + npg - 1) / npg);
    const size_t lds = ((MSG ? (size_t)sp.hin * kProjOut : 0) + 4 * kH + (sp.pd_lds ? (size_t)sp.N * kPdStride : 0)) * sizeof(float);
    hipLaunchKernelGGL((mpn_step_kernel<RE, MSG, MX>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool RE, bool MSG>
static hipError_t launch_step(const StepParams& sp, hipStream_t st) {
    return sp.agg == GNNCCA_AGG_MAX ? launch_step_t<RE, MSG, true>(sp, st) : launch_step_t<RE, MSG, false>(sp, st);
}


// ============================================================================================================
// Generic family: any legal GRAPH_NET_PARAMS outside the MFMA family (node latent != 32, edge latent != 6,
// multi-layer edge / node MLPs, deeper classifiers ...).  None of the shipped configs needs it; it exists so
// that the module is a drop-in for the whole constructor contract (models/mpn.py:154-247).  It follows the
// reference op for op -- virtual concatenation, Linear(+folded BN)(+ReLU) layer by layer, aggregation over the
// CSR segments in the caller's edge order (the order torch's CPU index_add_ sums in) -- with plain one-thread-
// per-output kernels: correctness first, no tuning.
// ============================================================================================================
struct GenSeg {
    const float* ptr;   // [rows][ld]
    const int* idx;     // optional row gather (row32 / col32 in the caller's edge order), or null
    int ld, width;
};

// out[r][o] = [ReLU](b[o] + sum over the concatenated segments of W[o][:] . in[r][:])
__global__ __launch_bounds__(256) void gen_dense_kernel(GenSeg s0, GenSeg s1, GenSeg s2, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ out, long long M,
                                                        int K, int O, int ld_out, int relu) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * O) return;
    const long long r = t / O;
    const int o = (int)(t - r * O);
    const float* __restrict__ w = W + (size_t)o * K;
    float acc = b[o];
    const GenSeg segs[3] = {s0, s1, s2};
    int koff = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const GenSeg& sg = segs[q];
        if (sg.width == 0) continue;
        const long long rr = sg.idx ? (long long)sg.idx[r] : r;
        const float* __restrict__ src = sg.ptr + (size_t)rr * sg.ld;
        for (int k = 0; k < sg.width; ++k) acc = fmaf(w[koff + k], src[k], acc);
        koff += sg.width;
    }
    out[(size_t)r * ld_out + o] = relu ? fmaxf(acc, 0.f) : acc;
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

// h[i][c] = agg over the segment of node i of m[k][c], k in the caller's edge order (models/mpn.py:99,192-202)
__global__ __launch_bounds__(256) void gen_aggregate_kernel(const float* __restrict__ m, const int* __restrict__ seg_ptr,
                                                            const int* __restrict__ perm, const unsigned* __restrict__ flags,
                                                            float* __restrict__ h, int N, int H, int agg) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)N * H) return;
    const int i = (int)(t / H), c = (int)(t - (long long)i * H);
    const unsigned fl = flags[0];
    if (fl & GNNCCA_GRAPH_BAD_INDEX) {
        h[t] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (fl & GNNCCA_GRAPH_UNSORTED) != 0;
    const int s = seg_ptr[i], e = seg_ptr[i + 1];
    float v = agg == GNNCCA_AGG_MAX ? -INFINITY : 0.f;
    for (int p = s; p < e; ++p) {
        const int k = unsorted ? perm[p] : p;
        const float x = m[(size_t)k * H + c];
        v = agg == GNNCCA_AGG_MAX ? fmaxf(v, x) : v + x;
    }
    if (agg == GNNCCA_AGG_MEAN) v = v / (float)max(e - s, 1);
    if (e == s) v = 0.f;
    h[t] = v;
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
            hipLaunchKernelGGL(gen_dense_kernel, grid1((size_t)total, 256), dim3(256), 0, st, a, b, c, blob + woff[l],
                               blob + boff[l], dst, M, L.in_dim, L.out_dim, ld, L.relu);
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
        hipLaunchKernelGGL(enc_gemm_plan_kernel, dim3((E + 255) / 256), dim3(256), 0, st, ep);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gen_index32_kernel, dim3((E + 255) / 256), dim3(256), 0, st, ei, E, N, row32o, col32o);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(gen_plan_finish_kernel, dim3(1), dim3(256), 0, st, ei, E, N, seg_ptr, col32, perm, cursor, flags,
                       (const unsigned*)blockflags);
    HIP_TRY(hipGetLastError());

    // encoder (models/mpn.py:270): node MLP on x, edge MLP on edge_attr
    int s;
    if (d->enc_node.n_layers > 0) {
        s = gen_run_mlp(d->enc_node, blob, hdr.w[0], hdr.b[0], GenSeg{x, nullptr, d->node_in, d->node_in}, none, none, N, h0, H,
                        nb[0], nb[1], nw, st);
        if (s != GNNCCA_OK) return s;
    } else {
        HIP_TRY(hipMemcpyAsync(h0, x, (size_t)N * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (trace && trace->h_enc) HIP_TRY(hipMemcpyAsync(trace->h_enc, h0, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
    if (E == 0) return GNNCCA_OK;
    if (d->enc_edge.n_layers > 0) {
        s = gen_run_mlp(d->enc_edge, blob, hdr.w[1], hdr.b[1], GenSeg{edge_attr, nullptr, d->edge_in, d->edge_in}, none, none, E,
                        e0, EF, eb[0], eb[1], ew, st);
        if (s != GNNCCA_OK) return s;
    } else {
        HIP_TRY(hipMemcpyAsync(e0, edge_attr, (size_t)E * EF * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (trace && trace->e_enc) HIP_TRY(hipMemcpyAsync(trace->e_enc, e0, (size_t)E * EF * 4, hipMemcpyDeviceToDevice, st));

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
            hipLaunchKernelGGL(gen_aggregate_kernel, grid1((size_t)N * H, 256), dim3(256), 0, st, (const float*)msg,
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

template <bool FIRST, bool CLS, bool MSG, bool PDL>
static hipError_t launch_fast(const StepParams& sp, hipStream_t st) {
    const int npg = 4 / sp.wps;
    const unsigned blocks = (unsigned)((sp.N + npg - 1) / npg);
    const size_t lds = ((MSG ? (size_t)kH * kProjOut : 0) + 4 * kH + (PDL ? (size_t)sp.N * kPdStride : 0)) * sizeof(float);
    hipLaunchKernelGGL((mpn_step_fast_kernel<FIRST, CLS, MSG, PDL>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

static hipError_t launch_fast_dispatch(const StepParams& sp, bool msg, hipStream_t st) {
    const int key = (sp.first ? 8 : 0) | (sp.cls_layers ? 4 : 0) | (msg ? 2 : 0) | (sp.pd_lds ? 1 : 0);
    switch (key) {
#define GNNCCA_FAST_CASE(K, A, B, C, D) \
    case K: return launch_fast<A, B, C, D>(sp, st);
        GNNCCA_FAST_CASE(0, false, false, false, false)
        GNNCCA_FAST_CASE(1, false, false, false, true)
        GNNCCA_FAST_CASE(2, false, false, true, false)
        GNNCCA_FAST_CASE(3, false, false, true, true)
        GNNCCA_FAST_CASE(4, false, true, false, false)
        GNNCCA_FAST_CASE(5, false, true, false, true)
        GNNCCA_FAST_CASE(6, false, true, true, false)
        GNNCCA_FAST_CASE(7, false, true, true, true)
        GNNCCA_FAST_CASE(8, true, false, false, false)
        GNNCCA_FAST_CASE(9, true, false, false, true)
        GNNCCA_FAST_CASE(10, true, false, true, false)
        GNNCCA_FAST_CASE(11, true, false, true, true)
        GNNCCA_FAST_CASE(12, true, true, false, false)
        GNNCCA_FAST_CASE(13, true, true, false, true)
        GNNCCA_FAST_CASE(14, true, true, true, false)
        GNNCCA_FAST_CASE(15, true, true, true, true)
#undef GNNCCA_FAST_CASE
    }
    return hipErrorInvalidValue;
}

}  // namespace gnncca

using namespace gnncca;

extern "C" {

#ifdef GNNCCA_STAMPS
__attribute__((visibility("default"))) int gnncca_debug_set_stamps(void* dev_buf) {
    unsigned long long* p = static_cast<unsigned long long*>(dev_buf);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)));
    return GNNCCA_OK;
}
#endif

int gnncca_last_hip_error(void) { return g_last_hip_error; }

int gnncca_read_graph_flags(const void* workspace, uint32_t* flags_out, gnncca_stream_t stream) {
    if (!workspace || !flags_out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(flags_out, workspace, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GNNCCA_OK;
}

}  // extern "C"

static int forward_impl(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                        const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                        float* logits_out, const gnncca_trace* trace, gnncca_stream_t stream, Profiler* prof) {
    if (!dims_valid(d) || n_nodes < 0 || n_edges < 0) return GNNCCA_ERR_INVALID_ARG;
    const Family fam = classify(d);
    if (fam == kFamilyNone) return GNNCCA_ERR_UNSUPPORTED;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    if (n_nodes == 0) return n_edges == 0 ? GNNCCA_OK : GNNCCA_ERR_INVALID_ARG;
    if (!packed_dev || !x || !workspace) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges > 0 && (!edge_index || !edge_attr || !logits_out)) return GNNCCA_ERR_INVALID_ARG;
    if (fam == kFamilyGeneric)
        return forward_generic(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes,
                               logits_out, trace, static_cast<hipStream_t>(stream));
    const Workspace ws = carve(d, n_nodes, n_edges);
    if (workspace_bytes < ws.total) return GNNCCA_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    const int N = (int)n_nodes, E = (int)n_edges;
    const float* blob = static_cast<const float*>(packed_dev);

    // The blob header is a pure function of dims: recompute it on the host instead of reading it back.
    BlobHeader hdr;
    if (!blob_header(d, &hdr)) return GNNCCA_ERR_UNSUPPORTED;

    unsigned* flags = reinterpret_cast<unsigned*>(base + ws.flags);
    int* seg_ptr = reinterpret_cast<int*>(base + ws.seg_ptr);
    int* col32 = reinterpret_cast<int*>(base + ws.col32);
    int* perm = reinterpret_cast<int*>(base + ws.perm);
    int* cursor = reinterpret_cast<int*>(base + ws.cursor);
    unsigned* blockflags = reinterpret_cast<unsigned*>(base + ws.blockflags);
    float* h0 = reinterpret_cast<float*>(base + ws.h0);
    float* act = reinterpret_cast<float*>(base + ws.act);
    float* part = reinterpret_cast<float*>(base + ws.partial);
    float* pd[2] = {reinterpret_cast<float*>(base + ws.pd[0]), reinterpret_cast<float*>(base + ws.pd[1])};
    float* psq[2] = {reinterpret_cast<float*>(base + ws.psq[0]), reinterpret_cast<float*>(base + ws.psq[1])};
    float* ebuf = reinterpret_cast<float*>(base + ws.e);
    float* e0buf = reinterpret_cast<float*>(base + ws.e0);

    // ---- node encoder -------------------------------------------------------------------------------------
    const int nl = d->enc_node.n_layers;
    const int n_gemm = nl == 1 ? 1 : nl - 1;
    const float* cur_in = x;
    int ks_last = 1;
    for (int g = 0; g < n_gemm; ++g) {
        const gnncca_layer& l = d->enc_node.layers[g];
        const int K = l.in_dim, O = l.out_dim;
        const int ks = g == 0 ? ws.ksplit : 1;
        int kslice = (K + ks - 1) / ks;
        kslice = (kslice + 63) / 64 * 64;
        const bool split = g == 0 && hdr.enc_w3 != 0 && N >= 4096 && (reinterpret_cast<uintptr_t>(cur_in) & 15) == 0;
        EncPlanParams ep;
        std::memset(&ep, 0, sizeof(ep));
        ep.in = cur_in;
        ep.W = blob + hdr.enc_node_w[g];
        ep.part = part;
        ep.M = N;
        ep.K = K;
        ep.O = O;
        ep.kslice = kslice;
        ep.vec_ok = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(cur_in) & 15) == 0);
        ep.nrt = (N + 31) / 32;
        ep.nks = ks;
        ep.gemm_blocks = split ? 0 : ep.nrt * ks * ((O + 127) / 128);
        int ks_split = 1;
        if (split) {  // big batches: split-bf16 MFMA GEMM; the plan gets its own launch
            const unsigned short* w3 = reinterpret_cast<const unsigned short*>(blob + hdr.enc_w3);
            if (N >= 32768) {  // 128-row workgroups: the weight tile is amortised over twice the rows
                hipLaunchKernelGGL((enc_gemm_split_kernel<64>), dim3((N + 127) / 128, 1), dim3(256), 0, st, cur_in, w3, part, N,
                                   K, O, K);
            } else {           // 64-row workgroups + split-K so that >= 512 workgroups are in flight
                while (ks_split < ws.ksplit && ((N + 63) / 64) * ks_split < 512 && (K / (ks_split * 2)) % 32 == 0) ks_split *= 2;
                hipLaunchKernelGGL((enc_gemm_split_kernel<32>), dim3((N + 63) / 64, ks_split), dim3(256), 0, st, cur_in, w3,
                                   part, N, K, O, K / ks_split);
            }
            HIP_TRY(hipGetLastError());
            PROF_MARK(GNNCCA_K_ENC_GEMM);
        }
        int plan_blocks = 0;
        if (g == 0 && E > 0) {  // the graph plan rides in the first GEMM launch
            ep.ei = reinterpret_cast<const long long*>(edge_index);
            ep.seg_ptr = seg_ptr;
            ep.col32 = col32;
            ep.blockflags = blockflags;
            ep.E = E;
            ep.N = N;
            plan_blocks = (E + 255) / 256;
        }
        if (ep.gemm_blocks + plan_blocks > 0) {
            hipLaunchKernelGGL(enc_gemm_plan_kernel, dim3(ep.gemm_blocks + plan_blocks), dim3(256), 0, st, ep);
            HIP_TRY(hipGetLastError());
            PROF_MARK(split ? GNNCCA_K_PLAN_ROWS : GNNCCA_K_ENC_GEMM);
        }
        ks_last = split ? ks_split : ks;
        if (g < n_gemm - 1) {
            float* dst = act + (size_t)(g & 1) * N * O;
            hipLaunchKernelGGL(reduce_bias_act_kernel, grid1((size_t)N * O, 256), dim3(256), 0, st, (const float*)part,
                               blob + hdr.enc_node_b[g], dst, N, O, ks, l.relu);
            HIP_TRY(hipGetLastError());
            PROF_MARK(GNNCCA_K_ENC_REDUCE);
            cur_in = dst;
        }
    }
    const int nf = d->reattach_nodes ? 2 : 1;
    const int hin = nf * kH;
    {
        const gnncca_layer& lprev = d->enc_node.layers[n_gemm - 1];
        TailParams tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.blob = blob;
        tp.part = part;
        tp.h0 = h0;
        tp.trace_h = trace ? trace->h_enc : nullptr;
        tp.pd_out = pd[0];
        tp.psq_out = psq[0];
        tp.off_prev_b = hdr.enc_node_b[n_gemm - 1];
        tp.off_lastWT = hdr.enc_last_wT;
        tp.off_last_b = hdr.enc_node_b[nl - 1];
        tp.off_projwT = hdr.proj_wT;
        tp.off_projb = hdr.proj_b;
        tp.ks = ks_last;
        tp.F = lprev.out_dim;
        tp.N = N;
        tp.has_last = nl >= 2;
        tp.relu_prev = lprev.relu;
        tp.reatt_n = d->reattach_nodes;
        tp.hin = hin;
        tp.vec_reduce = (tp.F % 4 == 0) && (tp.F / 4 <= 64) && (64 % (tp.F / 4) == 0);
        const size_t lds = ((size_t)hin * kProjOut + (tp.has_last ? (size_t)tp.F * kH : 0) + 4 * (size_t)tp.F + 4 * 256) * sizeof(float);
        if (lds > 160 * 1024) return GNNCCA_ERR_UNSUPPORTED;
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tail_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        tp.ei = reinterpret_cast<const long long*>(edge_index);
        tp.seg_ptr = seg_ptr;
        tp.col32 = col32;
        tp.perm = perm;
        tp.cursor = cursor;
        tp.flags = flags;
        tp.blockflags = blockflags;
        tp.E = E;
        const unsigned blocks = (unsigned)std::min<size_t>(((size_t)N + 3) / 4, 2048) + 1;  // + plan-repair workgroup
        hipLaunchKernelGGL(enc_tail_kernel, dim3(blocks), dim3(256), std::max<size_t>(lds, 4096), st, tp);
        HIP_TRY(hipGetLastError());
        PROF_MARK(GNNCCA_K_ENC_TAIL);
    }
    if (E == 0) return GNNCCA_OK;

    // ---- message passing steps ----------------------------------------------------------------------------
    const int L = d->num_enc_steps;
    const int first_cls = L - d->num_class_steps + 1;  // models/mpn.py:277
    const long long avg_deg = (E + (long long)N - 1) / N;
    const int chunks = (int)((avg_deg + 63) / 64);
    StepParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.blob = blob;
    sp.seg_ptr = seg_ptr;
    sp.col32 = col32;
    sp.perm = perm;
    sp.flags = flags;
    sp.edge_attr = edge_attr;
    sp.e = ebuf;
    sp.e0 = e0buf;
    sp.h0 = h0;
    sp.e_stride = ws.e_stride;
    sp.off_wee = hdr.wee;
    sp.off_wneb = hdr.wne_b;
    sp.off_projwT = hdr.proj_wT;
    sp.off_projb = hdr.proj_b;
    sp.off_encw = hdr.enc_edge_w;
    sp.off_encb = hdr.enc_edge_b;
    sp.off_cw1 = hdr.cls_w1;
    sp.off_cb1 = hdr.cls_b1;
    sp.off_cw2 = hdr.cls_w2;
    sp.off_cb2 = hdr.cls_b2;
    sp.off_fast = hdr.fast_consts;
    sp.cls_hidden = hdr.cls_hidden;
    sp.N = N;
    sp.E = E;
    sp.edge_in = d->edge_in;
    sp.attr_vec = d->edge_in == 4 && (reinterpret_cast<uintptr_t>(edge_attr) & 15) == 0;
    sp.agg = d->agg;
    sp.reatt_n = d->reattach_nodes;
    // waves per source-node segment: split a segment over 2 or 4 waves only while that is needed to put ~4 waves on
    // every SIMD (small graphs are latency-bound); big batches keep one wave per node, which amortises the
    // per-node prologue / projection epilogue over all of the node's chunks
    {
        const long long want = 4096 / (long long)N;  // waves per node that would fill the chip
        int wps = chunks >= 4 ? 4 : (chunks >= 2 ? 2 : 1);
        while (wps > 1 && wps > want) wps >>= 1;
        sp.wps = wps;
    }
    sp.hin = hin;
    sp.pd_lds = (N <= 1024) && d->num_enc_steps > 0;
    const bool re = d->reattach_edges != 0;
    int out_idx = 0;
    if (L == 0) {  // models/mpn.py:295-297: classify the encoded edge features once
        sp.first = 1;
        sp.update = 0;
        sp.cls_layers = hdr.cls_layers;
        sp.logits = logits_out;
        sp.trace_e_enc = trace ? trace->e_enc : nullptr;
        HIP_TRY(re ? (launch_step<true, false>(sp, st)) : (launch_step<false, false>(sp, st)));
        PROF_MARK(GNNCCA_K_STEP_LAST);
        return GNNCCA_OK;
    }
    for (int step = 1; step <= L; ++step) {
        const bool want_h = trace && trace->h_steps;
        const bool msg = step < L || want_h;
        sp.first = step == 1;
        sp.stamp_slot = 2 + (step - 1 < 6 ? step - 1 : 5);
        sp.update = 1;
        sp.store_e = step < L;
        sp.cls_layers = step >= first_cls ? hdr.cls_layers : 0;
        sp.logits = step >= first_cls ? logits_out + (size_t)(out_idx++) * E : nullptr;
        sp.pd_in = pd[(step - 1) & 1];
        sp.psq_in = psq[(step - 1) & 1];
        sp.pd_out = step < L ? pd[step & 1] : nullptr;
        sp.psq_out = step < L ? psq[step & 1] : nullptr;
        sp.trace_e_enc = (trace && step == 1) ? trace->e_enc : nullptr;
        sp.trace_e = (trace && trace->e_steps) ? trace->e_steps + (size_t)(step - 1) * E * kEF : nullptr;
        sp.trace_h = want_h ? trace->h_steps + (size_t)(step - 1) * N * kH : nullptr;
        hipError_t err;
        const bool fast = hdr.fast_consts != 0 && sp.attr_vec && !trace && d->agg != GNNCCA_AGG_MAX;
        if (fast)
            err = launch_fast_dispatch(sp, msg, st);
        else if (re)
            err = msg ? launch_step<true, true>(sp, st) : launch_step<true, false>(sp, st);
        else
            err = msg ? launch_step<false, true>(sp, st) : launch_step<false, false>(sp, st);
        HIP_TRY(err);
        PROF_MARK(msg ? GNNCCA_K_STEP : GNNCCA_K_STEP_LAST);
    }
    return GNNCCA_OK;
}

extern "C" {

int gnncca_mpn_forward(const gnncca_mpn_dims* d, const void* packed_dev, const float* x, const int64_t* edge_index,
                       const float* edge_attr, int64_t n_nodes, int64_t n_edges, void* workspace, size_t workspace_bytes,
                       float* logits_out, const gnncca_trace* trace, gnncca_stream_t stream) {
    return forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                        trace, stream, nullptr);
}

int gnncca_mpn_forward_profiled(const gnncca_mpn_dims* d, const void* packed_dev, const float* x,
                                const int64_t* edge_index, const float* edge_attr, int64_t n_nodes, int64_t n_edges,
                                void* workspace, size_t workspace_bytes, float* logits_out, gnncca_stream_t stream,
                                gnncca_profile* profile) {
    if (!profile) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Profiler p;
    p.out = profile;
    profile->count = 0;
    int s = prof_begin(&p, st);
    if (s != GNNCCA_OK) return s;
    s = forward_impl(d, packed_dev, x, edge_index, edge_attr, n_nodes, n_edges, workspace, workspace_bytes, logits_out,
                     nullptr, stream, &p);
    const int s2 = prof_end(&p, st);
    return s != GNNCCA_OK ? s : s2;
}

}  // extern "C"
