// graph_build.hip -- SURVEY.md 8f row N1 on gfx950: cross-camera graph construction + edge attributes for a batch of
// frames (replaces the per-frame Python / sklearn / GPU<->CPU round trips of inference.py:189-279).
//
// One wave per (SOURCE detection, slice of its frame's 64-candidate chunks), sources in the order the reference emits
// them (camera-major inside a frame).  A chunk is 64 consecutive detections of the frame, ONE PER LANE: the lane
// decides whether its candidate is on another camera (an edge, inference.py:210-211), and computes for it
//   ground-plane L2 and L1 distance in float64, divided by the frame's max_dist, cast to fp32   (inference.py:229-242)
//   the same-identity label                                                                     (inference.py:262-266)
// while the reid terms -- F.pairwise_distance(p=2, eps=1e-6), F.cosine_similarity(eps=1e-8)     (inference.py:222-226)
// -- are computed by the whole wave, four targets at a time: each target's reid row is one coalesced wave load, the
// 4 x 4 partial sums are reduced across the wave by a transposing butterfly (v_permlane32_swap, v_permlane16_swap,
// then 5 shuffles: 17 cross-lane ops for 16 sums instead of 96) and parked in the candidate's lane; sqrt / divide run
// once per chunk, lane-parallel.  The edge slot of a candidate is edge_ptr[source] + (other-camera candidates before
// it), a ballot prefix, so every store of edge_index / edge_attr / labels is a contiguous wave store in the
// reference's edge order.  Traffic: the reid table (N x R x 4 B) is re-read once per source from L2 -- it is 256 KB
// for a 256-detection frame -- and 8+8+16+4 B are written per edge; the kernel is L2/latency bound, not HBM bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

#include "internal.h"
#include "wave_reduce.cuh"

namespace gnncca {

#define HIP_TRY_GB(expr)                   \
    do {                                   \
        hipError_t _e = (expr);            \
        if (_e != hipSuccess) {            \
            g_last_hip_error = (int)_e;    \
            return GNNCCA_ERR_HIP;         \
        }                                  \
    } while (0)

typedef float f32x4g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void build_edges_kernel(const gnncca_frames fr, const float* __restrict__ reid, int R, int N,
                                                          long long E, long long* __restrict__ ei_out,
                                                          float* __restrict__ attr_out, float* __restrict__ lab_out,
                                                          int* __restrict__ zero_ptr, int zero_n) {
    constexpr int NA = MODE == GNNCCA_EDGE_ATTR_FULL ? 4 : 2;
    // (gnncca_frames_forward: the post-processing counters of the same batch are zeroed here, launches ahead of their first use, instead of
    // by a memset node of their own)
    if (zero_ptr && blockIdx.y == 0)
        for (int t = blockIdx.x * 256 + threadIdx.x; t < zero_n; t += gridDim.x * 256) zero_ptr[t] = 0;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= N) return;
    const int i = fr.src_order[p];
    const int g = fr.graph_of[i];
    const int gs = fr.graph_ptr[g], ge = fr.graph_ptr[g + 1];
    const int ci = fr.cam[i], pi = fr.person_id[i];
    const double xi = fr.xw[i], yi = fr.yw[i], md = fr.max_dist[g];
    const float* __restrict__ ri = reid + (size_t)i * R;
    const bool vec4 = (R & 3) == 0;
    long long pos = fr.edge_ptr[p];
    for (int c = 0, j0 = gs; j0 < ge; ++c, j0 += 64) {
        const int j = j0 + lane;
        const bool inb = j < ge;
        const bool valid = inb && fr.cam[j] != ci;  // same camera: no edge (inference.py:210-211)
        const unsigned long long mask = __ballot(valid);
        const int n_valid = __popcll(mask);
        if (c % (int)gridDim.y == (int)blockIdx.y && mask != 0ull) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, lab = 0.f;
            if (valid) {
                lab = fr.person_id[j] == pi ? 1.f : 0.f;
                if (MODE != GNNCCA_EDGE_ATTR_ONLY_APPEARANCE) {
                    // sklearn paired_distances on float64 rows, / max_dist, .type(float32); no FMA contraction
                    const double dx = __dsub_rn(xi, fr.xw[j]), dy = __dsub_rn(yi, fr.yw[j]);
                    const double l2 = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
                    const double l1 = __dadd_rn(fabs(dx), fabs(dy));
                    a0 = (float)__ddiv_rn(l2, md);
                    a1 = (float)__ddiv_rn(l1, md);
                }
            }
            if (MODE != GNNCCA_EDGE_ATTR_ONLY_DIST) {
                float sd = 0.f, sab = 0.f, saa = 1.f, sbb = 1.f;  // this lane's candidate: raw sums over the reid row
                unsigned long long m = mask;
                while (m != 0ull) {  // wave-uniform: four targets per round (the last round repeats its last target)
                    int t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (m != 0ull) {
                            t[k] = __ffsll((long long)m) - 1;
                            m &= m - 1;
                        } else {
                            t[k] = t[k - 1 < 0 ? 0 : k - 1];
                        }
                    }
                    const float* __restrict__ rj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) rj[k] = reid + (size_t)(j0 + t[k]) * R;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = 0.f;
                    auto acc = [&](int k, float a, float b) {
                        const float df = (a - b) + 1e-6f;  // F.pairwise_distance adds eps to the difference
                        v[4 * k + 0] = fmaf(df, df, v[4 * k + 0]);
                        v[4 * k + 1] = fmaf(a, b, v[4 * k + 1]);
                        v[4 * k + 2] = fmaf(a, a, v[4 * k + 2]);
                        v[4 * k + 3] = fmaf(b, b, v[4 * k + 3]);
                    };
                    if (vec4) {
                        for (int d = lane * 4; d < R; d += 256) {
                            const float4 a = *reinterpret_cast<const float4*>(ri + d);
                            float4 b[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) b[k] = *reinterpret_cast<const float4*>(rj[k] + d);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                acc(k, a.x, b[k].x);
                                acc(k, a.y, b[k].y);
                                acc(k, a.z, b[k].z);
                                acc(k, a.w, b[k].w);
                            }
                        }
                    } else {
                        for (int d = lane; d < R; d += 64) {
                            const float a = ri[d];
                            float b[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) b[k] = rj[k][d];
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc(k, a, b[k]);
                        }
                    }
                    const int toti = __float_as_int(transpose_reduce16(v));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float s0 = __int_as_float(__builtin_amdgcn_readlane(toti, lane_of_sum(4 * k + 0)));
                        const float s1 = __int_as_float(__builtin_amdgcn_readlane(toti, lane_of_sum(4 * k + 1)));
                        const float s2 = __int_as_float(__builtin_amdgcn_readlane(toti, lane_of_sum(4 * k + 2)));
                        const float s3 = __int_as_float(__builtin_amdgcn_readlane(toti, lane_of_sum(4 * k + 3)));
                        if (lane == t[k]) sd = s0, sab = s1, saa = s2, sbb = s3;
                    }
                }
                const float emb = sqrtf(sd);
                const float cosv = sab / (fmaxf(sqrtf(saa), 1e-8f) * fmaxf(sqrtf(sbb), 1e-8f));
                if (MODE == GNNCCA_EDGE_ATTR_FULL) {
                    a2 = emb;
                    a3 = cosv;
                } else {
                    a0 = emb;
                    a1 = cosv;
                }
            }
            if (valid) {
                const long long k = pos + __popcll(mask & ((1ull << lane) - 1ull));
                ei_out[k] = i;
                ei_out[E + k] = j;
                if (NA == 4) {
                    *reinterpret_cast<float4*>(attr_out + k * 4) = make_float4(a0, a1, a2, a3);
                } else {
                    *reinterpret_cast<float2*>(attr_out + k * 2) = make_float2(a0, a1);
                }
                lab_out[k] = lab;
            }
        }
        pos += n_valid;
    }
}

// Column norms of an [n_rows][n_cols] matrix (F.normalize(x, dim=0), inference.py:189-190), deterministic:
//   partial[chunk][c] = sum over the chunk's kColChunk rows of x[r][c]^2, rows in ascending order
//   norm[c]           = max(sqrt(sum over chunks, in four ordered quarters), 1e-12)
// The first version (256-row chunks, one dependent load per iteration, 512 workgroups) streamed the 2048-wide embedding
// matrix at 0.75 TB/s; here a thread owns four adjacent columns (16-B loads), eight rows are requested before they are
// squared and added (in order), and 64-row chunks put 4x as many workgroups on the chip.
constexpr int kColChunk = 64;
__global__ __launch_bounds__(256) void colnorm_partial_kernel(const float* __restrict__ x, long long n_rows, long long n_cols,
                                                              float* __restrict__ partial) {
    const long long r0 = (long long)blockIdx.y * kColChunk, r1 = min(r0 + kColChunk, n_rows);
    if ((n_cols & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long c = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (c >= n_cols) return;
        f32x4g s = {0.f, 0.f, 0.f, 0.f};
        for (long long r = r0; r < r1; r += 8) {
            f32x4g v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4g*>(x + min(r + u, r1 - 1) * n_cols + c);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + u < r1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] = fmaf(v[u][q], v[u][q], s[q]);
        }
        *reinterpret_cast<f32x4g*>(partial + (long long)blockIdx.y * n_cols + c) = s;
    } else {
        for (int q = 0; q < 4; ++q) {  // unaligned / odd widths: one column at a time
            const long long c = ((long long)blockIdx.x * 256 + threadIdx.x) * 4 + q;
            if (c >= n_cols) return;
            float s = 0.f;
            for (long long r = r0; r < r1; ++r) {
                const float v = x[r * n_cols + c];
                s = fmaf(v, v, s);
            }
            partial[(long long)blockIdx.y * n_cols + c] = s;
        }
    }
}

// 64 columns per workgroup: thread (quarter q = tid / 64, column = tid % 64) sums its quarter of the chunks in order,
// eight loads in flight; the four quarter sums are then added in order.
__global__ __launch_bounds__(256) void colnorm_finish_kernel(float* __restrict__ partial, long long n_chunks, long long n_cols) {
    __shared__ float s_q[4][64];
    const int q = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 64 + cl;
    const long long per = (n_chunks + 3) / 4, k0 = min((long long)q * per, n_chunks), k1 = min(k0 + per, n_chunks);
    float s = 0.f;
    if (c < n_cols) {
        for (long long k = k0; k < k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[min(k + u, k1 - 1) * n_cols + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < k1) s += v[u];
        }
    }
    s_q[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < n_cols) {
        const float t = ((s_q[0][cl] + s_q[1][cl]) + s_q[2][cl]) + s_q[3][cl];
        partial[n_chunks * n_cols + c] = fmaxf(sqrtf(t), 1e-12f);  // F.normalize clamps the norm at eps = 1e-12
    }
}

__global__ __launch_bounds__(256) void colnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                            long long total, long long n_cols, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) out[t] = x[t] / norm[t % n_cols];
}

// The three kernels above in ONE launch, for the matrices of a batch of frames (a few thousand rows: the launches, not the bytes, are
// what the per-batch pipeline pays for) and for up to two matrices at once (reid and node embeddings).  A workgroup owns 16 adjacent
// columns of one matrix (144 workgroups for 2048 + 256 columns): (1) chunk sums -- thread (chunk lane = tid / 4, four columns = tid % 4)
// runs colnorm_partial_kernel's fma chain over its 64-row chunks, sixteen rows requested before the first is used -> LDS;
// (2) colnorm_finish_kernel's four ordered quarter sums and the clamp; (3) the division, rows strided over the chunk lanes, eight in
// flight.  Same operations in the same order as the three-kernel form: the results are bit for bit the same.
// (First form: 32 columns per workgroup, eight rows in flight, one row per iteration in (3): 23.6 us for a 1229-row batch, a quarter of
// the Terrace pipeline's GPU time.)
constexpr int kFusedMaxChunks = 64;   // rows <= 4096
constexpr int kFusedCols = 16;
struct ColnormJob {
    const float* x;
    float* out;
    long long n_cols;
    int first_block;   // workgroups [first_block, first_block + ceil(n_cols / 16)) belong to this matrix
};
__global__ __launch_bounds__(256) void colnorm_fused_kernel(const ColnormJob j0, const ColnormJob j1, long long n_rows) {
    __shared__ float s_part[kFusedMaxChunks][kFusedCols];
    __shared__ float s_q[4][kFusedCols];
    __shared__ float s_norm[kFusedCols];
    const bool second = j1.x != nullptr && (int)blockIdx.x >= j1.first_block;
    const ColnormJob& j = second ? j1 : j0;
    const long long n_cols = j.n_cols;
    const float* __restrict__ x = j.x;
    const int tid = threadIdx.x, cl = tid >> 2, cg = tid & 3;
    const long long c0 = (long long)((int)blockIdx.x - j.first_block) * kFusedCols;
    const long long c = c0 + 4 * cg;
    const int n_chunks = (int)((n_rows + kColChunk - 1) / kColChunk);
    const bool vec = (n_cols & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(j.out) & 15) == 0;
    for (int ch = cl; ch < n_chunks; ch += 64) {
        const long long r0 = (long long)ch * kColChunk, r1 = min(r0 + kColChunk, n_rows);
        f32x4g s = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            if (c < n_cols)
                for (long long r = r0; r < r1; r += 16) {
                    f32x4g v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4g*>(x + min(r + u, r1 - 1) * n_cols + c);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (r + u < r1)
#pragma unroll
                            for (int q = 0; q < 4; ++q) s[q] = fmaf(v[u][q], v[u][q], s[q]);
                }
        } else {
            for (int q = 0; q < 4; ++q)
                if (c + q < n_cols)
                    for (long long r = r0; r < r1; ++r) {
                        const float v = x[r * n_cols + c + q];
                        s[q] = fmaf(v, v, s[q]);
                    }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s_part[ch][4 * cg + q] = s[q];
    }
    __syncthreads();
    if (tid < 4 * kFusedCols) {
        const int q = tid / kFusedCols, col = tid % kFusedCols;
        const int per = (n_chunks + 3) / 4, k0 = min(q * per, n_chunks), k1 = min(k0 + per, n_chunks);
        float t = 0.f;
        for (int k = k0; k < k1; ++k) t += s_part[k][col];
        s_q[q][col] = t;
    }
    __syncthreads();
    if (tid < kFusedCols) s_norm[tid] = fmaxf(sqrtf(((s_q[0][tid] + s_q[1][tid]) + s_q[2][tid]) + s_q[3][tid]), 1e-12f);
    __syncthreads();
    float* __restrict__ out = j.out;
    if (vec) {
        if (c < n_cols) {
            const f32x4g nv = {s_norm[4 * cg], s_norm[4 * cg + 1], s_norm[4 * cg + 2], s_norm[4 * cg + 3]};
            for (long long r = cl; r < n_rows; r += 64 * 8) {
                f32x4g v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4g*>(x + min(r + 64 * u, n_rows - 1) * n_cols + c);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r + 64 * u < n_rows)
                        *reinterpret_cast<f32x4g*>(out + (r + 64 * u) * n_cols + c) = f32x4g{v[u][0] / nv[0], v[u][1] / nv[1], v[u][2] / nv[2], v[u][3] / nv[3]};
            }
        }
    } else {
        for (int q = 0; q < 4; ++q)
            if (c + q < n_cols)
                for (long long r = cl; r < n_rows; r += 64) out[r * n_cols + c + q] = x[r * n_cols + c + q] / s_norm[4 * cg + q];
    }
}

// The same, for batches whose 16-column strip fits LDS (n_rows <= kLdsRowsMax: every batch of 64 Terrace frames): the strip is read from
// memory ONCE, by LDS-DMA (`buffer_load_dwordx4 ... lds`: 16 rows x 64 B per wave instruction, every request of the workgroup in flight
// at once, no VGPR staging), and both passes -- the chunk sums and the division -- run from LDS.  The kernel above walks its strip twice
// with 16 / 8 requests in flight per thread, and its first pass keeps 4 x ceil(rows / 64) of its 256 threads busy (72 for a 1100-row
// batch): 16.5 us for 10 MB in a pipeline whose kernels sum to 80.  Same operations in the same order: the same bits.
// LDS image: chunk ch (64 rows) at ch * kLdsChunkBytes, row-major [64][16] floats, 64 B of padding behind every chunk -- the chunk lanes
// of pass one then start 64 B apart modulo the 256-B bank row (no conflicts between the sixteen lanes of a ds_read_b128 pass).
constexpr int kLdsChunkBytes = kColChunk * kFusedCols * 4 + 64;
constexpr int kLdsRowsMax = 2304;   // 36 chunks: 149 760 B next to the 4.4 KB of static LDS
typedef __attribute__((address_space(3))) void* gb_lds_ptr;
__global__ __launch_bounds__(256) void colnorm_fused_lds_kernel(const ColnormJob j0, const ColnormJob j1, long long n_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_strip[];
    __shared__ float s_part[kFusedMaxChunks][kFusedCols];
    __shared__ float s_q[4][kFusedCols];
    __shared__ float s_norm[kFusedCols];
    const bool second = j1.x != nullptr && (int)blockIdx.x >= j1.first_block;
    const ColnormJob& j = second ? j1 : j0;
    const int n_cols = (int)j.n_cols;
    const int tid = threadIdx.x, lane = tid & 63, cl = tid >> 2, cg = tid & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = ((int)blockIdx.x - j.first_block) * kFusedCols;
    const int c = c0 + 4 * cg;
    const int rows = (int)n_rows;
    const int n_chunks = (rows + kColChunk - 1) / kColChunk;
    // (the host checked: 16-byte aligned matrices, n_cols % 4 == 0, n_rows * n_cols * 4 < 2^31; rows beyond the matrix read as zeros)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(j.x), 0, (int)((unsigned)rows * (unsigned)n_cols * 4u), 0x00020000);
    const unsigned col_ok = c < n_cols ? 0u : 0x80000000u;   // column groups beyond the matrix: out of range (zeros), never used
    const int n_groups = (rows + 15) / 16;
    for (int g = wave; g < n_groups; g += 4) {
        const unsigned voff = ((unsigned)(16 * g + (lane >> 2)) * (unsigned)n_cols + (unsigned)c) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (gb_lds_ptr)(s_strip + (g >> 2) * kLdsChunkBytes + (g & 3) * 1024), 16, voff | col_ok, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = cl; ch < n_chunks; ch += 64) {
        const int r0 = ch * kColChunk, r1 = min(r0 + kColChunk, rows);
        const unsigned char* base = s_strip + ch * kLdsChunkBytes + cg * 16;
        f32x4g s = {0.f, 0.f, 0.f, 0.f};
        for (int r = r0; r < r1; r += 16) {
            f32x4g v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4g*>(base + (min(r + u, r1 - 1) - r0) * (kFusedCols * 4));
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r + u < r1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] = fmaf(v[u][q], v[u][q], s[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s_part[ch][4 * cg + q] = s[q];
    }
    __syncthreads();
    if (tid < 4 * kFusedCols) {
        const int q = tid / kFusedCols, col = tid % kFusedCols;
        const int per = (n_chunks + 3) / 4, k0 = min(q * per, n_chunks), k1 = min(k0 + per, n_chunks);
        float t = 0.f;
        for (int k = k0; k < k1; ++k) t += s_part[k][col];
        s_q[q][col] = t;
    }
    __syncthreads();
    if (tid < kFusedCols) s_norm[tid] = fmaxf(sqrtf(((s_q[0][tid] + s_q[1][tid]) + s_q[2][tid]) + s_q[3][tid]), 1e-12f);
    __syncthreads();
    if (c < n_cols) {
        const f32x4g nv = {s_norm[4 * cg], s_norm[4 * cg + 1], s_norm[4 * cg + 2], s_norm[4 * cg + 3]};
        float* __restrict__ out = j.out;
        for (int r = cl; r < rows; r += 64) {   // (tid & 3 == cg for every piece of this thread: 256 % 4 == 0)
            const f32x4g v = *reinterpret_cast<const f32x4g*>(s_strip + (r >> 6) * kLdsChunkBytes + (r & 63) * (kFusedCols * 4) + cg * 16);
            *reinterpret_cast<f32x4g*>(out + (size_t)r * n_cols + c) = f32x4g{v[0] / nv[0], v[1] / nv[1], v[2] / nv[2], v[3] / nv[3]};
        }
    }
}

}  // namespace gnncca

using namespace gnncca;

extern "C" {

int gnncca_normalize_columns2(const float* x0, int64_t n_cols0, float* out0, const float* x1, int64_t n_cols1, float* out1, int64_t n_rows,
                              gnncca_stream_t stream) {
    if (n_rows < 0 || n_cols0 < 0 || n_cols1 < 0) return GNNCCA_ERR_INVALID_ARG;
    if ((n_rows + kColChunk - 1) / kColChunk > kFusedMaxChunks) return GNNCCA_ERR_UNSUPPORTED;
    if (n_rows == 0) return GNNCCA_OK;
    if ((n_cols0 > 0 && (!x0 || !out0)) || (n_cols1 > 0 && (!x1 || !out1))) return GNNCCA_ERR_INVALID_ARG;
    if (n_cols0 == 0) x0 = x1, out0 = out1, n_cols0 = n_cols1, n_cols1 = 0;
    if (n_cols0 == 0) return GNNCCA_OK;
    const long long b0 = (n_cols0 + kFusedCols - 1) / kFusedCols, b1 = (n_cols1 + kFusedCols - 1) / kFusedCols;
    if (b0 + b1 >= (1ll << 31)) return GNNCCA_ERR_UNSUPPORTED;
    ColnormJob j0{x0, out0, (long long)n_cols0, 0};
    ColnormJob j1{n_cols1 > 0 ? x1 : nullptr, out1, (long long)n_cols1, (int)b0};
    // the LDS-resident form where the strip fits and both matrices allow 16-byte accesses through 32-bit offsets
    auto vec_ok = [&](const float* x, const float* out, int64_t nc) {
        return nc == 0 || ((nc & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                           (long long)n_rows * nc * 4 < (1ll << 31));
    };
    static const bool no_lds = diag_env("GNNCCA_COLNORM_NOLDS") != nullptr;   // diagnostics: A/B against the two-pass form
    if (!no_lds && n_rows <= kLdsRowsMax && vec_ok(x0, out0, n_cols0) && vec_ok(x1, out1, n_cols1)) {
        const int lds = (int)((n_rows + kColChunk - 1) / kColChunk) * kLdsChunkBytes;
        static thread_local int attr_dev = -1;
        int dev = 0;
        HIP_TRY_GB(hipGetDevice(&dev));
        if (attr_dev != dev) {
            HIP_TRY_GB(hipFuncSetAttribute(reinterpret_cast<const void*>(colnorm_fused_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (kLdsRowsMax / kColChunk) * kLdsChunkBytes));
            attr_dev = dev;
        }
        hipLaunchKernelGGL(colnorm_fused_lds_kernel, dim3((unsigned)(b0 + b1)), dim3(256), (size_t)lds, static_cast<hipStream_t>(stream), j0, j1,
                           (long long)n_rows);
    } else {
        hipLaunchKernelGGL(colnorm_fused_kernel, dim3((unsigned)(b0 + b1)), dim3(256), 0, static_cast<hipStream_t>(stream), j0, j1, (long long)n_rows);
    }
    HIP_TRY_GB(hipGetLastError());
    return GNNCCA_OK;
}

int gnncca_normalize_columns(const float* x, int64_t n_rows, int64_t n_cols, float* scratch, float* out, gnncca_stream_t stream) {
    if (n_rows < 0 || n_cols < 0) return GNNCCA_ERR_INVALID_ARG;
    if (n_rows == 0 || n_cols == 0) return GNNCCA_OK;
    if (!x || !scratch || !out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long chunks = (n_rows + kColChunk - 1) / kColChunk;
    if (chunks > 65535) return GNNCCA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(colnorm_partial_kernel, dim3((unsigned)((n_cols + 1023) / 1024), (unsigned)chunks), dim3(256), 0, st, x,
                       (long long)n_rows, (long long)n_cols, scratch);
    HIP_TRY_GB(hipGetLastError());
    hipLaunchKernelGGL(colnorm_finish_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(256), 0, st, scratch, chunks,
                       (long long)n_cols);
    HIP_TRY_GB(hipGetLastError());
    const long long total = (long long)n_rows * n_cols;
    hipLaunchKernelGGL(colnorm_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x,
                       (const float*)(scratch + chunks * n_cols), total, (long long)n_cols, out);
    HIP_TRY_GB(hipGetLastError());
    return GNNCCA_OK;
}

int gnncca_build_edges(const gnncca_frames* fr, const float* reid, int32_t reid_dim, int64_t n_nodes, int64_t n_edges,
                       int32_t mode, int64_t* edge_index_out, float* edge_attr_out, float* edge_labels_out,
                       gnncca_stream_t stream) {
    return gnncca::build_edges_zeroing(fr, reid, reid_dim, n_nodes, n_edges, mode, edge_index_out, edge_attr_out, edge_labels_out, nullptr, 0, stream);
}

}  // extern "C"

int gnncca::build_edges_zeroing(const gnncca_frames* fr, const float* reid, int32_t reid_dim, int64_t n_nodes, int64_t n_edges,
                                int32_t mode, int64_t* edge_index_out, float* edge_attr_out, float* edge_labels_out, int32_t* zero_ptr,
                                int64_t zero_n, gnncca_stream_t stream) {
    if (!fr || n_nodes < 0 || n_edges < 0 || reid_dim < 0) return GNNCCA_ERR_INVALID_ARG;
    if (mode < GNNCCA_EDGE_ATTR_FULL || mode > GNNCCA_EDGE_ATTR_ONLY_DIST) return GNNCCA_ERR_INVALID_ARG;
    if (n_nodes == 0 || n_edges == 0) return GNNCCA_OK;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    if (!fr->xw || !fr->yw || !fr->max_dist || !fr->person_id || !fr->cam || !fr->graph_of || !fr->graph_ptr ||
        !fr->src_order || !fr->edge_ptr || !edge_index_out || !edge_attr_out || !edge_labels_out)
        return GNNCCA_ERR_INVALID_ARG;
    if (mode != GNNCCA_EDGE_ATTR_ONLY_DIST && (!reid || reid_dim == 0)) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // few sources: split each source's candidate chunks over up to 16 waves so that the launch still fills the chip
    unsigned slices = 1;
    while (slices < 16 && (long long)n_nodes * slices < 4096) slices *= 2;
    const dim3 grid((unsigned)((n_nodes + 3) / 4), slices), block(256);
    long long* ei = reinterpret_cast<long long*>(edge_index_out);
    switch (mode) {
        case GNNCCA_EDGE_ATTR_FULL:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_FULL>), grid, block, 0, st, *fr, reid, (int)reid_dim,
                               (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out, zero_ptr, (int)zero_n);
            break;
        case GNNCCA_EDGE_ATTR_ONLY_APPEARANCE:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_ONLY_APPEARANCE>), grid, block, 0, st, *fr, reid,
                               (int)reid_dim, (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out, zero_ptr, (int)zero_n);
            break;
        default:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_ONLY_DIST>), grid, block, 0, st, *fr, reid, (int)reid_dim,
                               (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out, zero_ptr, (int)zero_n);
            break;
    }
    HIP_TRY_GB(hipGetLastError());
    return GNNCCA_OK;
}

extern "C" {

// ---- host side of row N1: the edge enumeration of a batch of frames (inference.py:207-212), written straight into the staging image ----
// Layout of `staging` (what gnn_cca_amd.graph_build uploads in ONE transfer; 8-byte fields first):
//   f64 xw[n], yw[n], max_dist[g];  i64 ids[n];  i32 person[n], cam[n], graph_of[n], graph_ptr[g + 1], src_order[n], edge_ptr[n + 1],
//   edge_ptr_g[g + 1]
size_t gnncca_plan_frames_bytes(int64_t n, int64_t g) {
    if (n < 0 || g < 0) return 0;
    return (size_t)(8 * (3 * n + g) + 4 * (5 * n + 2 * g + 3));
}

int64_t gnncca_plan_frames(const double* xw, const double* yw, const int64_t* ids, const int64_t* id_cam, int64_t n,
                           const int64_t* graph_sizes, const double* max_dist, int64_t g, void* staging, size_t staging_bytes) {
    if (n < 0 || g < 0 || !staging || staging_bytes < gnncca_plan_frames_bytes(n, g)) return -(int64_t)GNNCCA_ERR_INVALID_ARG;
    if ((n > 0 && (!xw || !yw || !ids || !id_cam)) || (g > 0 && (!graph_sizes || !max_dist))) return -(int64_t)GNNCCA_ERR_INVALID_ARG;
    if (n >= (1ll << 31) - 64 || g >= (1ll << 31) - 64) return -(int64_t)GNNCCA_ERR_UNSUPPORTED;
    long long total = 0;
    for (int64_t q = 0; q < g; ++q) {
        if (graph_sizes[q] < 0) return -(int64_t)GNNCCA_ERR_INVALID_ARG;
        total += graph_sizes[q];
    }
    if (total != n) return -(int64_t)GNNCCA_ERR_INVALID_ARG;   // id_cam length does not match graph_sizes
    char* base = static_cast<char*>(staging);
    double* o_xw = reinterpret_cast<double*>(base);
    double* o_yw = o_xw + n;
    double* o_md = o_yw + n;
    int64_t* o_ids = reinterpret_cast<int64_t*>(o_md + g);
    int32_t* o_person = reinterpret_cast<int32_t*>(o_ids + n);
    int32_t* o_cam = o_person + n;
    int32_t* o_graph_of = o_cam + n;
    int32_t* o_graph_ptr = o_graph_of + n;
    int32_t* o_src = o_graph_ptr + g + 1;
    int32_t* o_edge_ptr = o_src + n;
    int32_t* o_edge_ptr_g = o_edge_ptr + n + 1;
    if (n) {
        std::memcpy(o_xw, xw, 8 * (size_t)n);
        std::memcpy(o_yw, yw, 8 * (size_t)n);
        std::memcpy(o_ids, ids, 8 * (size_t)n);
    }
    if (g) std::memcpy(o_md, max_dist, 8 * (size_t)g);
    // cameras: np.unique order (ascending camera id) inside every frame = the rank among the batch's distinct camera ids
    std::vector<int64_t> cams;
    for (int64_t i = 0; i < n; ++i) {
        if (id_cam[i] < INT32_MIN || id_cam[i] > INT32_MAX) return -(int64_t)GNNCCA_ERR_UNSUPPORTED;
        bool seen = false;
        for (int64_t c : cams) seen = seen || c == id_cam[i];
        if (!seen) {
            if (cams.size() >= 4096) return -(int64_t)GNNCCA_ERR_UNSUPPORTED;
            cams.push_back(id_cam[i]);
        }
    }
    std::sort(cams.begin(), cams.end());
    const int64_t n_cam = std::max<int64_t>((int64_t)cams.size(), 1);
    // person ids: any relabelling that preserves equality (gnncca_frames::person_id) -- first appearance, open addressing
    {
        size_t cap = 16;
        while (cap < (size_t)(2 * n + 1)) cap <<= 1;
        std::vector<int64_t> keys(cap);
        std::vector<int32_t> vals(cap, -1);
        int32_t next = 0;
        for (int64_t i = 0; i < n; ++i) {
            const uint64_t hsh = (uint64_t)ids[i] * 0x9E3779B97F4A7C15ull;
            size_t slot = (size_t)(hsh >> 20) & (cap - 1);
            while (vals[slot] >= 0 && keys[slot] != ids[i]) slot = (slot + 1) & (cap - 1);
            if (vals[slot] < 0) keys[slot] = ids[i], vals[slot] = next++;
            o_person[i] = vals[slot];
        }
    }
    // stable counting sort by (frame, camera rank): frame-major, camera order inside a frame, node id ascending inside a camera
    std::vector<int32_t> key((size_t)n);
    std::vector<int32_t> count((size_t)(g * n_cam) + 1, 0);
    {
        int64_t i = 0;
        o_graph_ptr[0] = 0;
        for (int64_t q = 0; q < g; ++q) {
            for (int64_t k = 0; k < graph_sizes[q]; ++k, ++i) {
                const int32_t rank = (int32_t)(std::lower_bound(cams.begin(), cams.end(), id_cam[i]) - cams.begin());
                o_graph_of[i] = (int32_t)q;
                o_cam[i] = (int32_t)id_cam[i];
                key[(size_t)i] = (int32_t)(q * n_cam + rank);
                ++count[(size_t)key[(size_t)i]];
            }
            o_graph_ptr[q + 1] = (int32_t)i;
        }
    }
    std::vector<int32_t> start((size_t)(g * n_cam) + 1, 0);
    for (size_t k = 0; k + 1 < start.size(); ++k) start[k + 1] = start[k] + count[k];
    {
        std::vector<int32_t> cursor(start);
        for (int64_t i = 0; i < n; ++i) o_src[cursor[(size_t)key[(size_t)i]]++] = (int32_t)i;
    }
    long long e = 0;
    for (int64_t pos = 0; pos < n; ++pos) {
        const int32_t node = o_src[pos];
        o_edge_ptr[pos] = (int32_t)e;
        e += graph_sizes[o_graph_of[node]] - count[(size_t)key[(size_t)node]];   // every node of the frame's OTHER cameras
        if (e >= (1ll << 31) - 64) return -(int64_t)GNNCCA_ERR_UNSUPPORTED;      // more than 2^31 edges in one batch
    }
    o_edge_ptr[n] = (int32_t)e;
    for (int64_t q = 0; q <= g; ++q) o_edge_ptr_g[q] = o_edge_ptr[o_graph_ptr[q]];   // edges are emitted frame by frame
    return (int64_t)e;
}

}  // extern "C"
