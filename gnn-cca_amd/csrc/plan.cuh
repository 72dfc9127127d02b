#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

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


}  // namespace gnncca
