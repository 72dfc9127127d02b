// post_host.cpp -- SURVEY.md 8f row N2, second half: the reference's bridge-based heuristics for ONE frame, on the host.
//
// The shipped inference configuration (config/config_inference.yaml:6-8) runs, after the threshold, PRUNING -> ROUNDING -> PRUNING ->
// SPLITTING (inference.py:306-345).  The device chain (csrc/postprocess.cuh) covers threshold, pruning, flow counts and clusters and raises
// two trigger bits per frame -- a node with flow > 3 (libs/utils.py:58-59), a cluster with more than four members (libs/utils.py:321);
// only frames that raise one need what is below, and SURVEY.md 8f leaves that on the CPU: graphs of <= 34 nodes, bridge searches, loops
// whose trip count depends on the data.  `gnncca_post_finalize_frame_host` is utils.compute_rounding (libs/utils.py:25-173),
// utils.remove_edges_single_direction (387-404), utils.disjoint_big_clusters (319-386) and utils.compute_SCC_and_Clusters (295-317) in
// that order, with the reference's OBSERVABLE behaviour -- its artefacts included, because they decide which edges go:
//   * cluster labels are positions in [strongly connected components in networkx's generation order, stably sorted by size] + [untouched
//     nodes in id order]; the splitting step works on the FIRST label with more than four members and re-reads "label l" in every
//     re-labelling -- so scc_generation_order() below follows networkx's algorithm (Nuutila's variant of Tarjan, nodes and successors in
//     insertion order), not just any SCC routine;
//   * rounding looks for bridges in the graph it was called with, every round (the reference never refreshes `predicted_active_edges`);
//   * splitting removes EVERY edge whose probability equals the minimum it found, looks for bridges in the whole frame and not in the big
//     cluster, and drops its recursive call's result except for that call's first in-place removal.
// Written from the reference's behaviour, checked against tests/golden/post2_heuristics.npz (the reference's own functions) and against
// oracle/post_oracle.py; plain C++17, no GPU, no networkx.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "internal.h"

namespace {

struct Frame {
    int n, E;
    const int64_t *src, *dst;   // frame-local ids after subtracting `base`
    int64_t base;
    const float* probs;
    std::vector<std::vector<int>> out_edges, in_edges;   // edge ids by source / by target node, ascending (index())
    int u(int k) const { return (int)(src[k] - base); }
    int v(int k) const { return (int)(dst[k] - base); }
    void index() {
        out_edges.assign(n, {}), in_edges.assign(n, {});
        for (int k = 0; k < E; ++k) out_edges[u(k)].push_back(k), in_edges[v(k)].push_back(k);
    }
};

typedef std::vector<int64_t> Pred;

// a set of ordered node pairs (u, v): a dense n x n byte map for frames (n <= 2048: 4 MB at most), a hash set beyond
struct PairSet {
    int n;
    std::vector<char> dense;
    std::unordered_set<uint64_t> sparse;
    explicit PairSet(int n_) : n(n_) {
        if (n <= 2048) dense.assign((size_t)n * n, 0);
    }
    bool insert(int u, int v) {   // true if new
        if (!dense.empty()) {
            char& c = dense[(size_t)u * n + v];
            const bool fresh = !c;
            c = 1;
            return fresh;
        }
        return sparse.insert((uint64_t)u * (uint64_t)n + (uint64_t)v).second;
    }
    bool has(int u, int v) const { return !dense.empty() ? dense[(size_t)u * n + v] != 0 : sparse.count((uint64_t)u * (uint64_t)n + (uint64_t)v) != 0; }
};

static std::vector<int> active_edges(const Frame& f, const Pred& p) {
    std::vector<int> a;
    for (int k = 0; k < f.E; ++k)
        if (p[k] == 1) a.push_back(k);
    return a;
}

// networkx.DiGraph(edge list): nodes in order of first appearance (u before v), successors in insertion order, parallel edges once
struct DiGraph {
    std::vector<int> nodes;               // insertion order
    std::vector<std::vector<int>> succ;   // indexed by node id (frame-local)
    std::vector<char> present;
};
static DiGraph digraph(const Frame& f, const std::vector<int>& act) {
    DiGraph g;
    g.succ.assign(f.n, {});
    g.present.assign(f.n, 0);
    PairSet seen(f.n);
    for (int k : act) {
        const int u = f.u(k), v = f.v(k);
        if (!g.present[u]) g.present[u] = 1, g.nodes.push_back(u);
        if (!g.present[v]) g.present[v] = 1, g.nodes.push_back(v);
        if (seen.insert(u, v)) g.succ[u].push_back(v);
    }
    return g;
}

// networkx.strongly_connected_components in generation order (networkx/algorithms/components/strongly_connected.py)
static std::vector<std::vector<int>> scc_generation_order(const DiGraph& g, int n) {
    std::vector<int> preorder(n, 0), lowlink(n, 0), it(n, 0);
    std::vector<char> found(n, 0);
    std::vector<int> sccq, queue;
    std::vector<std::vector<int>> out;
    int i = 0;
    for (int source : g.nodes) {
        if (found[source]) continue;
        queue.assign(1, source);
        while (!queue.empty()) {
            const int v = queue.back();
            if (!preorder[v]) preorder[v] = ++i;
            bool done = true;
            while (it[v] < (int)g.succ[v].size()) {          // the node's neighbour iterator keeps its position across visits
                const int w = g.succ[v][it[v]++];
                if (!preorder[w]) {
                    queue.push_back(w);
                    done = false;
                    break;
                }
            }
            if (!done) continue;
            lowlink[v] = preorder[v];
            for (int w : g.succ[v])
                if (!found[w]) lowlink[v] = std::min(lowlink[v], preorder[w] > preorder[v] ? lowlink[w] : preorder[w]);
            queue.pop_back();
            if (lowlink[v] == preorder[v]) {
                std::vector<int> scc(1, v);
                while (!sccq.empty() && preorder[sccq.back()] > preorder[v]) scc.push_back(sccq.back()), sccq.pop_back();
                for (int w : scc) found[w] = 1;
                out.push_back(std::move(scc));
            } else {
                sccq.push_back(v);
            }
        }
    }
    return out;
}

// utils.compute_SCC_and_Clusters: ids[v] = position of v's set in [SCCs stably sorted by size] + [untouched nodes in id order]
static int cluster_ids(const Frame& f, const std::vector<int>& act, std::vector<int>& ids) {
    const DiGraph g = digraph(f, act);
    std::vector<std::vector<int>> sets = scc_generation_order(g, f.n);
    std::stable_sort(sets.begin(), sets.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() < b.size(); });
    ids.assign(f.n, 0);
    int c = 0;
    for (const auto& s : sets) {
        for (int v : s) ids[v] = c;
        ++c;
    }
    for (int v = 0; v < f.n; ++v)
        if (!g.present[v]) ids[v] = c++;
    return c;
}

// both orientations of every bridge of the undirected graph of the active edges, as keys u * n + v
static std::unordered_set<uint64_t> bridge_set(const Frame& f, const std::vector<int>& act) {
    std::vector<std::vector<int>> und(f.n);
    {
        PairSet seen(f.n);
        for (int k : act) {
            const int u = f.u(k), v = f.v(k);
            if (u == v) continue;
            if (seen.insert(std::min(u, v), std::max(u, v))) und[u].push_back(v), und[v].push_back(u);
        }
    }
    std::vector<int> disc(f.n, 0), low(f.n, 0), it(f.n, 0), parent(f.n, -1), stack;
    std::unordered_set<uint64_t> out;
    int t = 0;
    for (int root = 0; root < f.n; ++root) {
        if (disc[root] || und[root].empty()) continue;
        disc[root] = low[root] = ++t;
        stack.assign(1, root);
        while (!stack.empty()) {
            const int v = stack.back();
            if (it[v] < (int)und[v].size()) {
                const int w = und[v][it[v]++];
                if (w == parent[v]) continue;                  // (simple graph: the one edge back to the parent)
                if (disc[w]) {
                    low[v] = std::min(low[v], disc[w]);
                } else {
                    parent[w] = v;
                    disc[w] = low[w] = ++t;
                    stack.push_back(w);
                }
            } else {
                stack.pop_back();
                const int p = parent[v];
                if (p >= 0) {
                    low[p] = std::min(low[p], low[v]);
                    if (low[v] > disc[p]) out.insert((uint64_t)p * f.n + v), out.insert((uint64_t)v * f.n + p);
                }
            }
        }
    }
    return out;
}

// utils.remove_edges_single_direction: an active edge survives iff its reverse is active too
static Pred prune(const Frame& f, const Pred& p) {
    PairSet act(f.n);
    for (int k = 0; k < f.E; ++k)
        if (p[k] == 1) act.insert(f.u(k), f.v(k));
    Pred out = p;
    for (int k = 0; k < f.E; ++k)
        if (p[k] == 1 && !act.has(f.v(k), f.u(k))) out[k] = 0;
    return out;
}

static void flows(const Frame& f, const Pred& p, std::vector<int>& fo, std::vector<int>& fi) {
    fo.assign(f.n, 0), fi.assign(f.n, 0);
    for (int k = 0; k < f.E; ++k)
        if (p[k] == 1) fo[f.u(k)]++, fi[f.v(k)]++;
}
static bool violated(const std::vector<int>& fo, const std::vector<int>& fi) {
    for (size_t v = 0; v < fo.size(); ++v)
        if (fo[v] > 3 || fi[v] > 3) return true;
    return false;
}

// utils.compute_rounding; false where the reference returns [] (no node with flow > 3: the caller keeps its predictions)
static bool rounding(const Frame& f, const Pred& pred, Pred& out) {
    std::vector<int> fo, fi;
    flows(f, pred, fo, fi);
    if (!violated(fo, fi)) return false;
    out = pred;
    const std::unordered_set<uint64_t> bridges = bridge_set(f, active_edges(f, pred));   // of the graph the call came with, every round
    std::vector<char> on_bridge(f.E, 0);
    if (!bridges.empty())
        for (int k = 0; k < f.E; ++k) on_bridge[k] = (char)bridges.count((uint64_t)f.u(k) * f.n + f.v(k));
    for (;;) {
        std::vector<int> remove;
        if (!bridges.empty()) {
            for (int side = 0; side < 2; ++side)
                for (int v = 0; v < f.n; ++v) {
                    if ((side == 0 ? fo[v] : fi[v]) <= 3) continue;
                    for (int k : (side == 0 ? f.out_edges[v] : f.in_edges[v]))   // ascending edge ids, as np.intersect1d returns them
                        if (out[k] == 1 && on_bridge[k]) remove.push_back(k);
                }
        }
        if (remove.empty()) {   // the weakest active edge of every violating node: first minimum in edge order (np.argmin)
            for (int side = 0; side < 2; ++side)
                for (int v = 0; v < f.n; ++v) {
                    if ((side == 0 ? fo[v] : fi[v]) <= 3) continue;
                    int best = -1;
                    for (int k : (side == 0 ? f.out_edges[v] : f.in_edges[v]))
                        if (out[k] == 1 && (best < 0 || f.probs[k] < f.probs[best])) best = k;
                    if (best >= 0) remove.push_back(best);
                }
        }
        for (int k : remove) out[k] = 0;
        flows(f, out, fo, fi);
        if (!violated(fo, fi)) return true;
    }
}

// utils.disjoint_big_clusters.  `pred` is the caller's object: modified in place exactly where the reference's tensor is; the returned
// vector is the reference's return value.
static Pred split_big_clusters(const Frame& f, std::vector<int> ids, Pred& pred, int depth) {
    int n_lab = 0;
    for (int v : ids) n_lab = std::max(n_lab, v + 1);
    std::vector<int> count(n_lab, 0);
    for (int v : ids) count[v]++;
    int lab = -1;
    for (int c = 0; c < n_lab; ++c)
        if (count[c] > 4) {
            lab = c;
            break;
        }
    if (lab < 0 || depth > f.E + 8) return pred;
    Pred* cur = &pred;
    Pred own;
    std::vector<int> act = active_edges(f, *cur);
    for (;;) {
        const std::unordered_set<uint64_t> bridges = bridge_set(f, act);
        float mn = 0.f;
        bool have = false;
        if (!bridges.empty()) {
            // predicted_act_edges.index(bridge): the FIRST active edge with that (u, v)
            std::unordered_map<uint64_t, int> first;
            for (int k : act) first.emplace((uint64_t)f.u(k) * f.n + f.v(k), k);
            for (uint64_t b : bridges) {
                auto q = first.find(b);
                if (q == first.end()) continue;   // (cannot happen after pruning: both directions are active)
                if (!have || f.probs[q->second] < mn) mn = f.probs[q->second], have = true;
            }
        } else {
            for (int k : act)
                if (ids[f.u(k)] == lab || ids[f.v(k)] == lab)
                    if (!have || f.probs[k] < mn) mn = f.probs[k], have = true;
        }
        if (!have) return *cur;   // nothing to remove: the reference would raise on an empty minimum; unreachable with a cluster of five
        for (int k = 0; k < f.E; ++k)
            if (f.probs[k] == mn) (*cur)[k] = 0;
        act = active_edges(f, *cur);
        cluster_ids(f, act, ids);
        int members = 0;
        for (int v : ids) members += v == lab;
        own = prune(f, *cur);     // a new object from here on
        cur = &own;
        act = active_edges(f, own);
        if (members <= 4) {
            (void)split_big_clusters(f, ids, own, depth + 1);   // its result is dropped; its first in-place removal stays in `own`
            return own;
        }
    }
}

}  // namespace

extern "C" {

int gnncca_post_finalize_frame_host(const int64_t* src, const int64_t* dst, int64_t node_base, int64_t n_nodes, int64_t n_edges,
                                    const float* probs, int64_t* predictions, int32_t switches, int32_t* labels_out,
                                    int32_t* n_clusters_out, int64_t* id_pred_out) {
    if (n_nodes < 0 || n_edges < 0 || n_nodes >= (1ll << 24) || n_edges >= (1ll << 30)) return GNNCCA_ERR_INVALID_ARG;
    if (n_edges > 0 && (!src || !dst || !probs || !predictions)) return GNNCCA_ERR_INVALID_ARG;
    Frame f{(int)n_nodes, (int)n_edges, src, dst, node_base, probs, {}, {}};
    for (int k = 0; k < f.E; ++k)
        if (src[k] < node_base || src[k] >= node_base + n_nodes || dst[k] < node_base || dst[k] >= node_base + n_nodes) return GNNCCA_ERR_INVALID_ARG;
    f.index();
    Pred pred(predictions, predictions + n_edges);
    const bool do_round = (switches & GNNCCA_POST_ROUNDING) != 0, do_prune = (switches & GNNCCA_POST_PRUNING) != 0,
               do_split = (switches & GNNCCA_POST_SPLITTING) != 0;
    if (do_prune) pred = prune(f, pred);
    if (do_round) {
        Pred r;
        if (rounding(f, pred, r)) pred = r;
    }
    if (do_prune) pred = prune(f, pred);
    std::vector<int> ids;
    int k = cluster_ids(f, active_edges(f, pred), ids);
    if (do_split) {
        Pred work = pred;
        pred = split_big_clusters(f, ids, work, 0);
        k = cluster_ids(f, active_edges(f, pred), ids);
    }
    for (int e = 0; e < f.E; ++e) predictions[e] = pred[e];
    if (id_pred_out)
        for (int v = 0; v < f.n; ++v) id_pred_out[v] = ids[v];
    if (labels_out) {   // the device chain's convention: the smallest (batch-global) node id of the component
        std::vector<int> smallest(k, f.n);
        for (int v = 0; v < f.n; ++v) smallest[ids[v]] = std::min(smallest[ids[v]], v);
        for (int v = 0; v < f.n; ++v) labels_out[v] = (int32_t)(smallest[ids[v]] + node_base);
    }
    if (n_clusters_out) *n_clusters_out = k;
    return GNNCCA_OK;
}

// The same for a list of frames of ONE batch (Batch.from_data_list layout: frame g owns nodes [node_ptr[g], node_ptr[g + 1]) and the contiguous
// edges [edge_ptr[g], edge_ptr[g + 1]); src / dst / probs / predictions / labels are the BATCH's arrays): frames[i] names a frame to finalize,
// clusters_out[i] receives its final cluster count; frames are independent, so they are dealt to up to `n_threads` host threads (0: one per
// hardware thread, at most 16).  Returns the first non-zero status of any frame.
int gnncca_post_finalize_frames_host(const int64_t* src, const int64_t* dst, const int32_t* node_ptr, const int32_t* edge_ptr,
                                     const int32_t* frames, int32_t n_listed, const float* probs, int64_t* predictions, int32_t switches,
                                     int32_t* labels, int32_t* clusters_out, int32_t n_threads) {
    if (n_listed < 0 || (n_listed > 0 && (!node_ptr || !edge_ptr || !frames || !clusters_out))) return GNNCCA_ERR_INVALID_ARG;
    if (n_listed == 0) return GNNCCA_OK;
    int nt = n_threads > 0 ? n_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::min(nt, (int)n_listed);
    std::atomic<int> next(0), status(GNNCCA_OK);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n_listed; i = next.fetch_add(1)) {
            const int g = frames[i];
            const int64_t v0 = node_ptr[g], v1 = node_ptr[g + 1], k0 = edge_ptr[g], k1 = edge_ptr[g + 1];
            const int st = gnncca_post_finalize_frame_host(src + k0, dst + k0, v0, v1 - v0, k1 - k0, probs + k0, predictions + k0, switches,
                                                           labels ? labels + v0 : nullptr, clusters_out + i, nullptr);
            if (st != GNNCCA_OK) {
                int expected = GNNCCA_OK;
                status.compare_exchange_strong(expected, st);
            }
        }
    };
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return status.load();
}

}  // extern "C"
