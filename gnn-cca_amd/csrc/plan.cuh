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

// Edges one plan workgroup (256 threads) covers: one per thread on small graphs (the plan rides in the latency-bound
// GEMM launch there), four per thread from 2^20 edges on (four independent 8-byte loads in flight per thread and a
// quarter of the workgroups: the thread-per-edge form was bound by per-workgroup latency, 2.8 TB/s).
__host__ __device__ inline int plan_edges_per_block(int E) { return E >= (1 << 20) ? 1024 : 256; }
__host__ __device__ inline int plan_num_blocks(int E) { return (E + plan_edges_per_block(E) - 1) / plan_edges_per_block(E); }

// Per-edge part of the plan.  Every workgroup reports its findings in its OWN word (`blockflags[b]`, always
// written, so there is no state to clear between forwards); the tail launch ORs them into flags[0].
__device__ __forceinline__ void plan_block(int pb, const long long* __restrict__ ei, int E, int N,
                                           int* __restrict__ seg_ptr, int* __restrict__ col32,
                                           unsigned* __restrict__ blockflags, unsigned* s_fl, int ell_S = 0) {
    if (threadIdx.x == 0) *s_fl = 0u;
    __syncthreads();
    const int per = plan_edges_per_block(E) / 256;
    const int k0 = pb * 256 * per + threadIdx.x;
    long long r[4], c[4], rp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // all loads first
        const int k = k0 + u * 256;
        const bool on = u < per && k < E;
        r[u] = on ? ei[k] : 0;
        c[u] = on ? ei[(size_t)E + k] : 0;
        rp[u] = (on && k > 0) ? ei[k - 1] : -1;
    }
    unsigned fl = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 256;
        if (!(u < per && k < E)) continue;
        if (r[u] < 0 || r[u] >= N || c[u] < 0 || c[u] >= N) {
            fl |= GNNCCA_GRAPH_BAD_INDEX;
            continue;
        }
        col32[k] = (int)c[u];
        const bool prev_ok = k == 0 || (rp[u] >= 0 && rp[u] < N);  // otherwise its owner raises the flag
        if (prev_ok) {
            if (r[u] < rp[u]) {
                fl |= GNNCCA_GRAPH_UNSORTED;
            } else {
                for (long long n = rp[u] + 1; n <= r[u]; ++n) seg_ptr[n] = k;
                // padded layout of the step kernels (ell_S slots per node, chosen from E/N): a row that still continues
                // ell_S edges after its start does not fit (rows are sorted here, or UNSORTED is raised elsewhere)
                if (ell_S > 0 && r[u] > rp[u] && (long long)k + ell_S < E && ei[(size_t)k + ell_S] == r[u])
                    fl |= GNNCCA_GRAPH_IRREGULAR;
            }
            if (k == E - 1)
                for (long long n = r[u] + 1; n <= N; ++n) seg_ptr[n] = E;
        }
    }
    if (fl) atomicOr(s_fl, fl);
    __syncthreads();
    if (threadIdx.x == 0) blockflags[pb] = *s_fl;
}

// OR of the per-block findings -> flags[0]; stable counting sort if the rows were not sorted.  `smem` >= 3 KB.
__device__ void plan_finish(const long long* __restrict__ ei, int E, int N, int* seg_ptr, int* col32, int* perm,
                            int* cursor, unsigned* flags, const unsigned* __restrict__ blockflags, unsigned* smem) {
    const int tid = threadIdx.x;
    unsigned fl = 0u;
    const int nb = plan_num_blocks(E);
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
