#pragma once
// the step right after the MPN on the GPU -- sigmoid / threshold (inference.py:286-291), pruning
// of edges that are active in one direction only (libs/utils.py:387-404), per-node flow counts (libs/utils.py:54-59)
// and the identity clusters of the pruned, now symmetric, edge set (libs/utils.py:295-317: strongly connected
// components of a symmetric digraph == connected components).  The bridge-based rounding / splitting heuristics
// (libs/utils.py:25-173, 319-386) stay on the host, as SURVEY.md 8f prescribes.
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

__global__ __launch_bounds__(256) void post_threshold_kernel(const float* __restrict__ logits, long long E,
                                                             float* __restrict__ probs, long long* __restrict__ preds) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    const float p = 1.f / (1.f + expf(-logits[k]));  // torch.nn.Sigmoid
    probs[k] = p;
    preds[k] = p >= 0.5f ? 1 : 0;                    // (preds_prob >= 0.5) * 1
}

// out[k] = pred[k] && the reverse edge (col k, row k) exists and is active too.  The reverse edge is searched in the
// CSR segment of node col[k] (plan of the MPN forward: seg_ptr / col32 / perm), any column order.
// FUSED_THRESHOLD (gnncca_frames_forward): `logits` instead of `pred` -- the thread computes its edge's probability and prediction with
// post_threshold_kernel's own expression (same bits), writes both out, and judges a reverse edge by that expression too: one launch less.
template <bool FUSED_THRESHOLD>
__global__ __launch_bounds__(256) void post_prune_kernel(const long long* __restrict__ ei, const long long* __restrict__ pred,
                                                         long long E, const int* __restrict__ seg_ptr,
                                                         const int* __restrict__ col32, const int* __restrict__ perm,
                                                         const unsigned* __restrict__ flags, long long* __restrict__ out,
                                                         int* __restrict__ flow_out, int* __restrict__ flow_in,
                                                         const float* __restrict__ logits, float* __restrict__ probs_out,
                                                         long long* __restrict__ preds_out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    const unsigned fl = flags[0];
    auto active = [&](long long q) -> bool {
        if (FUSED_THRESHOLD) return 1.f / (1.f + expf(-logits[q])) >= 0.5f;
        return pred[q] == 1;
    };
    bool mine;
    if (FUSED_THRESHOLD) {
        const float p = 1.f / (1.f + expf(-logits[k]));  // torch.nn.Sigmoid (post_threshold_kernel)
        probs_out[k] = p;
        mine = p >= 0.5f;
        preds_out[k] = mine ? 1 : 0;
    } else {
        mine = pred[k] == 1;
    }
    if (fl & GNNCCA_GRAPH_BAD_INDEX) {
        out[k] = 0;
        return;
    }
    long long keep = 0;
    if (mine) {
        const bool unsorted = (fl & GNNCCA_GRAPH_UNSORTED) != 0;
        const int i = (int)ei[k], j = (int)ei[E + k];
        for (int q = seg_ptr[j]; q < seg_ptr[j + 1]; ++q) {
            if (col32[q] == i && active(unsorted ? perm[q] : q)) {
                keep = 1;
                break;
            }
        }
        // the flow counts of the pruned edge set (libs/utils.py:54-59) in the same pass (a launch of its own until round 4);
        // the indices are in range here (no BAD_INDEX).  Integer counts: order-independent.
        if (keep && flow_out) {
            atomicAdd(&flow_out[i], 1);
            atomicAdd(&flow_in[j], 1);
        }
    }
    out[k] = keep;
}

// Connected components of the active edges by hooking + pointer jumping inside one workgroup.  labels[v] = smallest
// node id of v's component.  Without frame ranges the whole batch is one workgroup's job; with them (node_ptr /
// edge_ptr of a disjoint union of frame graphs, edges of a frame contiguous: Batch.from_data_list's layout) every frame
// gets its own workgroup -- components never cross frames -- and the batch is processed frames-wide in parallel.
// *n_clusters must be zero on entry.
// `triggers` (optional; zero on entry, with `sizes` [N_all]): this workgroup's frame sets bit 0 of its word when one of its nodes has
// flow_out or flow_in > 3 (libs/utils.py:58-62: compute_rounding has work to do) and bit 1 when one of its clusters has more than four
// members (libs/utils.py:321-322: disjoint_big_clusters has) -- the two conditions under which the host heuristics change the frame.
// LABEL CONVENTION (relied on by the host pass: csrc/post_host.cpp counts the clusters the device chain found in a flagged frame as the nodes with
// labels[v] == v, and writes its own results in the same form): a node's label is the SMALLEST batch-global node id of its component.
__global__ __launch_bounds__(1024) void post_cc_kernel(const long long* __restrict__ ei, const long long* __restrict__ pred,
                                                       long long E_all, int N_all, const int* __restrict__ node_ptr,
                                                       const int* __restrict__ edge_ptr, int* labels, int* n_clusters,
                                                       const int* __restrict__ flow_out, const int* __restrict__ flow_in, int* sizes,
                                                       int* triggers) {
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    const int v0 = node_ptr ? node_ptr[blockIdx.x] : 0, v1 = node_ptr ? node_ptr[blockIdx.x + 1] : N_all;
    const long long k0 = edge_ptr ? edge_ptr[blockIdx.x] : 0, k1 = edge_ptr ? edge_ptr[blockIdx.x + 1] : E_all;
    const long long E = E_all;
    const int N = v1 - v0;  // bound on the number of sweeps
    auto ld = [&](int v) { return __hip_atomic_load(&labels[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int v = v0 + tid; v < v1; v += 1024) __hip_atomic_store(&labels[v], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int round = 0; round <= N; ++round) {  // terminates when a sweep hooks nothing; N bounds it
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (long long k = k0 + tid; k < k1; k += 1024) {
            if (pred[k] != 1) continue;
            const long long a = ei[k], b = ei[E + k];
            if (a < v0 || a >= v1 || b < v0 || b >= v1) continue;  // out of range, or an edge that leaves its frame: ignored
            int ra = (int)a, rb = (int)b;
            for (int p = ld(ra); p != ra; p = ld(ra)) ra = p;  // find roots
            for (int p = ld(rb); p != rb; p = ld(rb)) rb = p;
            if (ra != rb) {
                atomicMin(&labels[max(ra, rb)], min(ra, rb));   // hook the larger root under the smaller
                s_changed = 1;
            }
        }
        __syncthreads();
        for (int v = v0 + tid; v < v1; v += 1024) {             // pointer jumping
            int r = v;
            for (int p = ld(r); p != r; p = ld(r)) r = p;
            __hip_atomic_store(&labels[v], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_changed) break;
        __syncthreads();
    }
    int mine = 0;
    for (int v = v0 + tid; v < v1; v += 1024) mine += ld(v) == v;
    if (mine) atomicAdd(n_clusters, mine);
    if (triggers) {
        int trig = 0;
        for (int v = v0 + tid; v < v1; v += 1024) {
            if (flow_out[v] > 3 || flow_in[v] > 3) trig |= 1;      // (final: the prune kernel is a launch of its own, before this one)
            atomicAdd(&sizes[ld(v)], 1);
        }
        __syncthreads();   // (the adds above are counted in vmcnt, which the barrier waits for: they are performed)
        for (int v = v0 + tid; v < v1; v += 1024)
            if (ld(v) == v && __hip_atomic_load(&sizes[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4) trig |= 2;
        if (trig) atomicOr(&triggers[blockIdx.x], trig);
    }
}

// ---- frame padding for the one-graph-fits-all-frames replay (gnn_cca_amd.inference.GraphedForward(pad_to=...)) ---------------------------
// The per-frame loop of inference.py:173-283 sees another (N, E) every frame; a HIP graph is captured for ONE shape.  Independent components of
// a disjoint union do not influence each other (Batch.from_data_list semantics, inference.py:279), so a frame can be padded to a canonical
// shape with a DUMMY component: `n_dummy` extra nodes behind the real ones that carry every padding edge as a self loop (rows stay sorted: the
// dummy ids are the largest; the loops are dealt over the dummy nodes in order so that no single segment holds them all).  One launch copies
// the frame into the padded buffers and writes the padding: x rows beyond N are zeroed, padding edges get zero attributes.
__global__ __launch_bounds__(256) void pad_frame_kernel(const float* __restrict__ x, long long n, const long long* __restrict__ ei,
                                                        const float* __restrict__ ea, long long e, float* __restrict__ x_pad,
                                                        long long n_real_max, int n_dummy, long long* __restrict__ ei_pad,
                                                        float* __restrict__ ea_pad, long long e_pad, int node_in, int edge_in) {
    const long long n_pad = n_real_max + n_dummy;
    const long long nx = n_pad * node_in, na = e_pad * edge_in;
    const long long total = nx + 2 * e_pad + na;
    const long long per = (e_pad - e + n_dummy - 1) / max(n_dummy, 1);   // padding loops per dummy node
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        if (t < nx) {
            const long long r = t / node_in;
            x_pad[t] = r < n ? x[t] : 0.f;
        } else if (t < nx + 2 * e_pad) {
            const long long u = t - nx, side = u / e_pad, k = u - side * e_pad;
            ei_pad[u] = k < e ? ei[side * e + k] : n_real_max + (per > 0 ? (k - e) / per : 0);
        } else {
            const long long u = t - nx - 2 * e_pad, k = u / edge_in;
            ea_pad[u] = k < e ? ea[u] : 0.f;
        }
    }
}

}  // namespace gnncca
