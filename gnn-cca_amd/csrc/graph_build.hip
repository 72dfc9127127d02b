// graph_build.hip -- SURVEY.md 8f row N1 on gfx950: cross-camera graph construction + edge attributes for a batch of
// frames (replaces the per-frame Python / sklearn / GPU<->CPU round trips of inference.py:189-279).
//
// One wave per SOURCE detection, in the order the reference emits sources (camera-major inside a frame).  The wave
// walks the frame's detections in ascending id, skips its own camera, and for every target computes
//   ground-plane L2 and L1 distance in float64, divided by the frame's max_dist, cast to fp32   (inference.py:229-242)
//   F.pairwise_distance(p=2, eps=1e-6) and F.cosine_similarity(eps=1e-8) of the reid rows        (inference.py:222-226)
//   the same-identity label                                                                     (inference.py:262-266)
// with the reid row of the target read as one coalesced 1 KB wave load and reduced across the wave by DPP shuffles.
// Results are parked one per lane and flushed 64 at a time, so every store of edge_index / edge_attr / labels is a
// contiguous wave store.  Traffic: the reid table (N x R x 4 B) is re-read once per source from L2 -- it is 256 KB for
// a 256-detection frame -- and 8+8+16+4 B are written per edge; the kernel is L2/latency bound, not HBM bound.
#include <hip/hip_runtime.h>

#include "internal.h"

namespace gnncca {

#define HIP_TRY_GB(expr)                   \
    do {                                   \
        hipError_t _e = (expr);            \
        if (_e != hipSuccess) {            \
            g_last_hip_error = (int)_e;    \
            return GNNCCA_ERR_HIP;         \
        }                                  \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void build_edges_kernel(const gnncca_frames fr, const float* __restrict__ reid, int R, int N,
                                                          long long E, long long* __restrict__ ei_out,
                                                          float* __restrict__ attr_out, float* __restrict__ lab_out) {
    constexpr int NA = MODE == GNNCCA_EDGE_ATTR_FULL ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= N) return;
    const int i = fr.src_order[p];
    const int g = fr.graph_of[i];
    const int gs = fr.graph_ptr[g], ge = fr.graph_ptr[g + 1];
    const int ci = fr.cam[i], pi = fr.person_id[i];
    const double xi = fr.xw[i], yi = fr.yw[i], md = fr.max_dist[g];
    const float* __restrict__ ri = reid + (size_t)i * R;
    long long pos = fr.edge_ptr[p];
    // parked results of up to 64 targets (lane q holds target number q of the current batch)
    int pj = 0;
    float pa[4] = {0.f, 0.f, 0.f, 0.f};
    float pl = 0.f;
    int parked = 0;
    auto flush = [&]() {
        if (lane < parked) {
            const long long k = pos + lane;
            ei_out[k] = i;
            ei_out[E + k] = pj;
            if (NA == 4) {
                *reinterpret_cast<float4*>(attr_out + k * 4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
            } else {
                *reinterpret_cast<float2*>(attr_out + k * 2) = make_float2(pa[0], pa[1]);
            }
            lab_out[k] = pl;
        }
        pos += parked;
        parked = 0;
    };
    for (int j = gs; j < ge; ++j) {
        if (fr.cam[j] == ci) continue;  // wave-uniform: same camera, no edge (inference.py:210-211)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (MODE != GNNCCA_EDGE_ATTR_ONLY_APPEARANCE) {
            // sklearn paired_distances on float64 rows, / max_dist, .type(float32); no FMA contraction
            const double dx = __dsub_rn(xi, fr.xw[j]), dy = __dsub_rn(yi, fr.yw[j]);
            const double l2 = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            const double l1 = __dadd_rn(fabs(dx), fabs(dy));
            a0 = (float)__ddiv_rn(l2, md);
            a1 = (float)__ddiv_rn(l1, md);
        }
        if (MODE != GNNCCA_EDGE_ATTR_ONLY_DIST) {
            const float* __restrict__ rj = reid + (size_t)j * R;
            float sd = 0.f, sab = 0.f, saa = 0.f, sbb = 0.f;
            for (int d = lane; d < R; d += 64) {
                const float a = ri[d], b = rj[d];
                const float df = (a - b) + 1e-6f;  // F.pairwise_distance adds eps to the difference
                sd = fmaf(df, df, sd);
                sab = fmaf(a, b, sab);
                saa = fmaf(a, a, saa);
                sbb = fmaf(b, b, sbb);
            }
            sd = wave_sum(sd);
            sab = wave_sum(sab);
            saa = wave_sum(saa);
            sbb = wave_sum(sbb);
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
        const float lab = fr.person_id[j] == pi ? 1.f : 0.f;
        if (lane == parked) {
            pj = j;
            pa[0] = a0, pa[1] = a1, pa[2] = a2, pa[3] = a3;
            pl = lab;
        }
        if (++parked == 64) flush();
    }
    flush();
}

// partial[chunk][c] = sum over the chunk's 256 rows of x[r][c]^2, rows in ascending order
__global__ __launch_bounds__(256) void colnorm_partial_kernel(const float* __restrict__ x, long long n_rows, long long n_cols,
                                                              float* __restrict__ partial) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    const long long r0 = (long long)blockIdx.y * 256, r1 = min(r0 + 256, n_rows);
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) {
        const float v = x[r * n_cols + c];
        s = fmaf(v, v, s);
    }
    partial[(long long)blockIdx.y * n_cols + c] = s;
}

__global__ __launch_bounds__(256) void colnorm_finish_kernel(float* __restrict__ partial, long long n_chunks, long long n_cols) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    float s = 0.f;
    for (long long k = 0; k < n_chunks; ++k) s += partial[k * n_cols + c];
    partial[n_chunks * n_cols + c] = fmaxf(sqrtf(s), 1e-12f);  // F.normalize clamps the norm at eps = 1e-12
}

__global__ __launch_bounds__(256) void colnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                            long long total, long long n_cols, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) out[t] = x[t] / norm[t % n_cols];
}

}  // namespace gnncca

using namespace gnncca;

extern "C" {

int gnncca_normalize_columns(const float* x, int64_t n_rows, int64_t n_cols, float* scratch, float* out, gnncca_stream_t stream) {
    if (n_rows < 0 || n_cols < 0) return GNNCCA_ERR_INVALID_ARG;
    if (n_rows == 0 || n_cols == 0) return GNNCCA_OK;
    if (!x || !scratch || !out) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long chunks = (n_rows + 255) / 256;
    const unsigned cb = (unsigned)((n_cols + 255) / 256);
    if (chunks > 65535) return GNNCCA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(colnorm_partial_kernel, dim3(cb, (unsigned)chunks), dim3(256), 0, st, x, (long long)n_rows,
                       (long long)n_cols, scratch);
    HIP_TRY_GB(hipGetLastError());
    hipLaunchKernelGGL(colnorm_finish_kernel, dim3(cb), dim3(256), 0, st, scratch, chunks, (long long)n_cols);
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
    if (!fr || n_nodes < 0 || n_edges < 0 || reid_dim < 0) return GNNCCA_ERR_INVALID_ARG;
    if (mode < GNNCCA_EDGE_ATTR_FULL || mode > GNNCCA_EDGE_ATTR_ONLY_DIST) return GNNCCA_ERR_INVALID_ARG;
    if (n_nodes == 0 || n_edges == 0) return GNNCCA_OK;
    if (n_nodes >= (1ll << 31) - 64 || n_edges >= (1ll << 31) - 64) return GNNCCA_ERR_UNSUPPORTED;
    if (!fr->xw || !fr->yw || !fr->max_dist || !fr->person_id || !fr->cam || !fr->graph_of || !fr->graph_ptr ||
        !fr->src_order || !fr->edge_ptr || !edge_index_out || !edge_attr_out || !edge_labels_out)
        return GNNCCA_ERR_INVALID_ARG;
    if (mode != GNNCCA_EDGE_ATTR_ONLY_DIST && (!reid || reid_dim == 0)) return GNNCCA_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n_nodes + 3) / 4)), block(256);
    long long* ei = reinterpret_cast<long long*>(edge_index_out);
    switch (mode) {
        case GNNCCA_EDGE_ATTR_FULL:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_FULL>), grid, block, 0, st, *fr, reid, (int)reid_dim,
                               (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out);
            break;
        case GNNCCA_EDGE_ATTR_ONLY_APPEARANCE:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_ONLY_APPEARANCE>), grid, block, 0, st, *fr, reid,
                               (int)reid_dim, (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out);
            break;
        default:
            hipLaunchKernelGGL((build_edges_kernel<GNNCCA_EDGE_ATTR_ONLY_DIST>), grid, block, 0, st, *fr, reid, (int)reid_dim,
                               (int)n_nodes, (long long)n_edges, ei, edge_attr_out, edge_labels_out);
            break;
    }
    HIP_TRY_GB(hipGetLastError());
    return GNNCCA_OK;
}

}  // extern "C"
