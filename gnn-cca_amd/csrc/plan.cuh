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
// GEMM launch there), four per thread from 2^19 edges on (a quarter of the workgroups and, with aligned rows, the pair form of
// plan_block: the thread-per-edge form was bound by per-workgroup latency, 2.8 TB/s).  Round 3 lowered the threshold from 2^20:
// the per-GPU share of BASELINE config 4 (64 x dense128, E = 1 040 384) sat just below it.
__host__ __device__ inline int plan_edges_per_block(int E) { return E >= (1 << 19) ? 1024 : 256; }
__host__ __device__ inline int plan_num_blocks(int E) { return (E + plan_edges_per_block(E) - 1) / plan_edges_per_block(E); }
// The plan launch of big batches takes plan_block's pair form when the index rows allow 16-byte loads: kPlanSpan 1024-edge blocks
// per workgroup.  (Measured with 4 blocks = 4096 edges per workgroup, every load in flight before the first use: 64 x dense256
// 19.5 -> 20.7 us, 512 x dense128 39.2 -> 42.7, 64 x dense128 6.9 -> 8.5, profiles/r03_logs/r3_plan_ab2.log; with the padded-layout
// check's row ids requested up front, two blocks: 5.7 / 18.6 / 34.7 / 51.3 -> 6.3 / 18.8 / 33.8 / 48 us at 64 x dense128 / 64 x dense256 /
// 512 x dense128 / 200 x dense256 on a box 4 % slower, r3_plan_ahead1.log vs r3_plan_ahead2.log: a tie; one block it is.)
constexpr int kPlanSpan = 1;
inline int plan_span(const void* ei, int E) {   // 0: the narrow form
    return (plan_edges_per_block(E) == 1024 && (E & 1) == 0 && (reinterpret_cast<unsigned long long>(ei) & 15) == 0) ? kPlanSpan : 0;
}

// One edge of the plan (shared by both forms below): bounds, int32 target id, CSR offsets at row boundaries, padded-layout check.
// `ahead`: the row id of edge k + ell_S when the caller has it in a register already (HAVE_AHEAD; the pair form requests it with
// its other loads, so that a row start does not cost the workgroup a second, dependent round trip), else it is read here.
template <bool HAVE_AHEAD = false>
__device__ __forceinline__ unsigned plan_edge(int k, long long r, long long c, long long rp, const long long* __restrict__ ei, int E, int N,
                                              int* __restrict__ seg_ptr, int& col_out, int ell_S, long long ahead = 0) {
    if (r < 0 || r >= N || c < 0 || c >= N) return GNNCCA_GRAPH_BAD_INDEX;
    unsigned fl = 0u;
    col_out = (int)c;
    const bool prev_ok = k == 0 || (rp >= 0 && rp < N);  // otherwise its owner raises the flag
    if (prev_ok) {
        if (r < rp) {
            fl |= GNNCCA_GRAPH_UNSORTED;
        } else {
            for (long long n = rp + 1; n <= r; ++n) seg_ptr[n] = k;
            // padded layout of the step kernels (ell_S slots per node, chosen from E/N): a row that still continues
            // ell_S edges after its start does not fit (rows are sorted here, or UNSORTED is raised elsewhere)
            if (ell_S > 0 && r > rp && (long long)k + ell_S < E && (HAVE_AHEAD ? ahead : ei[(size_t)k + ell_S]) == r) fl |= GNNCCA_GRAPH_IRREGULAR;
        }
        if (k == E - 1)
            for (long long n = r + 1; n <= N; ++n) seg_ptr[n] = E;
    }
    return fl;
}

// Per-edge part of the plan.  Every workgroup reports its findings in its OWN word (`blockflags[b]`, always
// written, so there is no state to clear between forwards); the tail launch ORs them into flags[0].
// `tx`: the thread's index inside ITS 256-thread plan block (default threadIdx.x).  A 512-thread workgroup runs two plan blocks side by side
// (enc_f16_slices.cuh: tx = threadIdx.x & 255, one `s_fl` word per half): both halves pass the same barriers.
__device__ __forceinline__ void plan_block(int pb, const long long* __restrict__ ei, int E, int N,
                                           int* __restrict__ seg_ptr, int* __restrict__ col32,
                                           unsigned* __restrict__ blockflags, unsigned* s_fl, int ell_S = 0, int span = 0, int tx = -1) {
    if (tx < 0) tx = (int)threadIdx.x;
    if (tx == 0) *s_fl = 0u;
    __syncthreads();
    const int per = plan_edges_per_block(E) / 256;
    unsigned fl = 0u;
    // Batches (1024 edges per workgroup), both index rows 16-byte aligned: a thread owns PAIRS of consecutive edges -- one 16-byte
    // non-temporal load per row and pair (edge_index is read once per forward), all four in flight before the first use, the
    // previous edge's row id from the neighbouring lane instead of a third load, the two narrowed ids as one 8-byte store.
    // (Round 2's form read every row id twice, 8 bytes at a time: 3.7-4.2 TB/s of the 20 B / edge.)
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    if (span == kPlanSpan) {   // (the host checked: per == 4, E even, both index rows 16-byte aligned)
        const int lane = tx & 63;
        constexpr int U = 2 * kPlanSpan;
        ll2 r2[U], c2[U], a2[U];
        long long rlast[U];
        int kk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = pb * (1024 * kPlanSpan) + u * 512 + 2 * tx;
            kk[u] = k;
            const bool on = k < E;   // E is even: a pair is inside or outside as a whole
            const int kl = on ? k : 0;
            r2[u] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(ei + kl));
            c2[u] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(ei + (size_t)E + kl));
            // the edge before a wave's first pair belongs to another wave (or workgroup): lane 0 fetches it
            rlast[u] = (lane == 0 && on && k > 0) ? ei[k - 1] : -1;
            // padded-layout check (plan_edge): the row ids ell_S edges ahead, requested NOW with everything else -- ell_S is a multiple
            // of 32, so this is an aligned pair too, and a line another wave streams anyway (L2, not HBM) -- instead of by the row-start
            // lanes after the first round trip (a second, dependent one for the whole workgroup: 64 x dense256 19.5-20 -> 18.5 us, 512 x dense128 37-39 -> 34.7, 64 x dense128
            // 6.4 -> 5.7; profiles/r03_logs/r3_plan_ahead1.log)
            a2[u] = (ell_S > 0 && (long long)k + ell_S < E) ? *reinterpret_cast<const ll2*>(ei + (size_t)k + ell_S) : ll2{-1, -1};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kk[u];
            const long long up = __shfl_up(r2[u][1], 1);
            const long long rp0 = lane == 0 ? rlast[u] : up;
            if (k < E) {
                int j0 = 0, j1 = 0;
                const unsigned f0 = plan_edge<true>(k, r2[u][0], c2[u][0], rp0, ei, E, N, seg_ptr, j0, ell_S, a2[u][0]);
                const unsigned f1 = plan_edge<true>(k + 1, r2[u][1], c2[u][1], r2[u][0], ei, E, N, seg_ptr, j1, ell_S, a2[u][1]);
                fl |= f0 | f1;
                if (!((f0 | f1) & GNNCCA_GRAPH_BAD_INDEX)) {
                    *reinterpret_cast<i32x2*>(col32 + k) = i32x2{j0, j1};
                } else {   // the forward is poisoned anyway; keep what is valid as the narrow form does
                    if (!(f0 & GNNCCA_GRAPH_BAD_INDEX)) col32[k] = j0;
                    if (!(f1 & GNNCCA_GRAPH_BAD_INDEX)) col32[k + 1] = j1;
                }
            }
        }
    } else {
        const int k0 = pb * 256 * per + tx;
        long long r[4], c[4], rp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // all loads first
            const int k = k0 + u * 256;
            const bool on = u < per && k < E;
            r[u] = on ? ei[k] : 0;
            c[u] = on ? ei[(size_t)E + k] : 0;
            rp[u] = (on && k > 0) ? ei[k - 1] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 256;
            if (!(u < per && k < E)) continue;
            int j = 0;
            const unsigned f = plan_edge(k, r[u], c[u], rp[u], ei, E, N, seg_ptr, j, ell_S);
            fl |= f;
            if (!(f & GNNCCA_GRAPH_BAD_INDEX)) col32[k] = j;
        }
    }
    if (fl) atomicOr(s_fl, fl);
    __syncthreads();
    // one word per 1024-edge (or 256-edge) block, always written: a wide workgroup reports its findings in each of its blocks' words
    const int nb = plan_num_blocks(E), wspan = span > 1 ? span : 1;
    if (tx < wspan && pb * wspan + tx < nb) blockflags[pb * wspan + tx] = *s_fl;
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
    if (tid == 0) flags[0] = fl, flags[1] = 0u;   // flags[1]: raised by step 1 when a node's columns are not <= 2 contiguous ranges
    if ((fl & GNNCCA_GRAPH_UNSORTED) && !(fl & GNNCCA_GRAPH_BAD_INDEX)) {
        int* si = reinterpret_cast<int*>(smem);
        plan_sort_fallback(ei, E, N, seg_ptr, col32, perm, cursor, si, si + 256, si + 512);
    }
}


}  // namespace gnncca
