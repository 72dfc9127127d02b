#pragma once
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

// ============================================================================================================
// Generic family, eval forward, ONE launch per message-passing step (round 4; rounds 1-3 ran the reference's ops one launch each:
// ~35 launches, 0.26-0.29 ms for a dense-256 forward off the shipped widths).
//
// The structure of the MFMA family's step kernels with run-time widths and VALU arithmetic:
//   * a workgroup owns one source node's CSR segment (models/mpn.py:99 aggregates by `row`), a thread owns an edge of the current
//     tile of T edges;
//   * the first Linear of the edge MLP is split by the cat order of models/mpn.py:68, [x[row] | x[col] | e]: the x[row] and x[col]
//     parts are per-NODE projections (P_src incl. the bias, P_dst), computed once per node by the previous step's epilogue (step 1:
//     gen_project_kernel) -- per edge only the e part (ein_w columns) is left, plus a gather of P_dst[col]; likewise the first
//     Linear of the node MLP, [x[row] | e'] (mpn.py:97): Q = W_x h + b per node, W_e e' per edge;
//   * every further layer (edge MLP, node MLP, classifier) is a small dense layer on the thread's activations, which live in LDS
//     k-major ([width][T + 1] floats: conflict-free for "thread = edge" and for the channel-major reduction below); weights are read
//     with wave-uniform addresses straight from the blob (transposed and padded [in][ceil8(out)], four outputs per pass);
//   * aggregation: thread (channel c, part p) sums its part of the tile's T messages of channel c from LDS into a register that
//     lives across the tiles of the segment; parts are combined in fixed order at the end (sum / mean / max; empty segment -> 0);
//   * epilogue: h'[node] -> HBM, then the next step's projections of cat(h0, h') or h' (reattach_initial_nodes, mpn.py:285).
// Edges are processed in the plan's sorted order and addressed in the CALLER's order through `perm` when the rows were not
// sorted, as everywhere in the generic family.  Results differ from the op-by-op path in summation order only.
// ============================================================================================================
struct GenLayerDesc {
    int woff, boff, in, out, op, relu;   // Wt at blob + woff: [in][op]; b at blob + boff: [op]
};
struct GenMlpDesc {
    int n;
    GenLayerDesc l[GNNCCA_MAX_LAYERS];
};
struct GenStepParams {
    const float* blob;
    const int* seg_ptr;
    const int* col32;
    const int* perm;
    const unsigned* flags;
    int N, E;
    const float* e_a;   // first block of the edge input: e0 when reattaching edges (mpn.py:283), else the latent edge features
    const float* e_b;   // second block (the latent when reattaching), or null
    int e_a_ld, e_a_w, e_b_ld, e_b_w;
    const float* tab_in;    // [N][tab_ld]: P_src (+ bias) | P_dst | Q (+ bias) of this step
    float* tab_out;         // the next step's, or null
    int tab_ld, o1e, o1n;   // widths of the first edge / node layer
    GenMlpDesc edge, node, cls;
    int k0_edge, k0_node;   // first column of the e block in the first layer's weight (2 * hin_w, hin_w)
    float* e_new;           // [E][e_new_ld] latent edge features after this step (caller's edge order)
    int e_new_ld;
    float* logits;          // [E] or null: this step does not classify
    float* h_new;           // [N][H] or null: the node update is not needed (last step)
    const float* h0;        // [N][H]: initial node features (reattach_nodes), else null
    int H, EF, hin_w, agg;
    int lds_stride;         // T + 1
    int wmax;               // widest activation vector a thread keeps in LDS (gen_fused_ok)
};

// one dense layer on the tile: s_out[o][t] = act(init(o) + sum_k Wt[k0 + k][o] * s_in[k][t]), four outputs per pass
template <typename Init>
__device__ __forceinline__ void gen_layer_lds(const float* __restrict__ blob, const GenLayerDesc& L, int k0, int K, const float* s_in,
                                              float* s_out, int TS, int t, Init init) {
    const float* __restrict__ Wt = blob + L.woff + (size_t)k0 * L.op;
    for (int o0 = 0; o0 < L.out; o0 += 4) {
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = init(o0 + j);
        for (int k = 0; k < K; ++k) {
            const float x = s_in[k * TS + t];
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wt + (size_t)k * L.op + o0);   // wave-uniform address; op is a multiple of 8
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[j], x, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (o0 + j < L.out) s_out[(o0 + j) * TS + t] = L.relu ? fmaxf(acc[j], 0.f) : acc[j];
    }
}

// the per-node projections of the NEXT step from hin = cat(h0, h) or h, all threads of the workgroup; s_hin: [hin_w] in LDS
__device__ __forceinline__ void gen_project_node(const GenStepParams& p, const float* s_hin, int node) {
    const GenLayerDesc& Le = p.edge.l[0];
    const GenLayerDesc& Ln = p.node.l[0];
    const int total = 2 * p.o1e + p.o1n;
    float* __restrict__ dst = p.tab_out + (size_t)node * p.tab_ld;
    for (int s = threadIdx.x; s < total; s += blockDim.x) {
        const float* __restrict__ Wt;
        int op, o;
        float acc;
        if (s < p.o1e) {                 // P_src = W[:, 0:hin] hin + b
            o = s, op = Le.op, Wt = p.blob + Le.woff, acc = p.blob[Le.boff + o];
        } else if (s < 2 * p.o1e) {      // P_dst = W[:, hin:2 hin] hin
            o = s - p.o1e, op = Le.op, Wt = p.blob + Le.woff + (size_t)p.hin_w * op, acc = 0.f;
        } else {                         // Q = W_n[:, 0:hin] hin + b_n
            o = s - 2 * p.o1e, op = Ln.op, Wt = p.blob + Ln.woff, acc = p.blob[Ln.boff + o];
        }
        for (int k = 0; k < p.hin_w; ++k) acc = fmaf(Wt[(size_t)k * op + o], s_hin[k], acc);
        dst[s] = acc;
    }
}

// step 1's tables: one workgroup per node, hin from HBM
__global__ __launch_bounds__(256) void gen_project_kernel(const GenStepParams p, const float* __restrict__ h_cur) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int node = blockIdx.x;
    for (int k = threadIdx.x; k < p.hin_w; k += blockDim.x)
        smem[k] = p.h0 ? (k < p.H ? p.h0[(size_t)node * p.H + k] : h_cur[(size_t)node * p.H + k - p.H]) : h_cur[(size_t)node * p.H + k];
    __syncthreads();
    gen_project_node(p, smem, node);
}

__global__ __launch_bounds__(256) void gen_step_fused_kernel(const GenStepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = blockDim.x, TS = p.lds_stride, t = threadIdx.x;
    const int node = blockIdx.x;
    // LDS: two activation buffers [wmax][TS], then [hin_w] + [parts][H] scratch of the epilogue
    const int wmax = p.wmax;
    float* s_a = smem;
    float* s_b = s_a + (size_t)wmax * TS;
    float* s_x = s_b + (size_t)wmax * TS;       // [hin_w] then [parts][H]
    const unsigned fl = p.flags[0];
    if (fl & GNNCCA_GRAPH_BAD_INDEX) {   // the plan is not trustworthy: poison what this workgroup would have written, touch nothing else
        if (p.logits)
            for (size_t k = (size_t)blockIdx.x * T + t; k < (size_t)p.E; k += (size_t)gridDim.x * T) p.logits[k] = __builtin_nanf("");
        if (p.h_new)
            for (int c = t; c < p.H; c += T) p.h_new[(size_t)node * p.H + c] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (fl & GNNCCA_GRAPH_UNSORTED) != 0;
    const int seg_s = p.seg_ptr[node], seg_t = p.seg_ptr[node + 1];
    const float* __restrict__ tab = p.tab_in + (size_t)node * p.tab_ld;   // this node's P_src | . | Q (wave-uniform reads)
    const int ein_w = p.e_a_w + p.e_b_w;
    // aggregation roles: thread (channel c, part pt) for c < H; parts split the T edges of a tile evenly
    const int parts = max(1, T / max(p.H, 1));
    const int chunk = (T + parts - 1) / parts;
    const int rc = t % max(p.H, 1), rp = t / max(p.H, 1);
    const bool reducer = p.h_new != nullptr && t < p.H * parts;
    const float ident = p.agg == GNNCCA_AGG_MAX ? -INFINITY : 0.f;
    float hacc = ident;
    for (int base = seg_s; base < seg_t; base += T) {
        const int pos = base + t;
        const bool live = pos < seg_t;
        const int pc = live ? pos : seg_t - 1;
        const int k = unsorted ? p.perm[pc] : pc;      // the caller's id of this thread's edge
        const int j = p.col32[pc];                      // its target node
        // ---- edge MLP (models/mpn.py:68-69): inputs cat(e0, e) or e into s_a, k-major --------------------------------------------
        for (int q = 0; q < p.e_a_w; ++q) s_a[q * TS + t] = p.e_a[(size_t)k * p.e_a_ld + q];
        for (int q = 0; q < p.e_b_w; ++q) s_a[(p.e_a_w + q) * TS + t] = p.e_b[(size_t)k * p.e_b_ld + q];
        const float* __restrict__ pdst = p.tab_in + (size_t)j * p.tab_ld + p.o1e;
        float* cur = s_a;
        float* nxt = s_b;
        gen_layer_lds(p.blob, p.edge.l[0], p.k0_edge, ein_w, cur, nxt, TS, t,
                      [&](int o) { return o < p.o1e ? tab[o] + pdst[o] : 0.f; });
        {
            float* tmp = cur;
            cur = nxt, nxt = tmp;
        }
        for (int l = 1; l < p.edge.n; ++l) {
            const GenLayerDesc& L = p.edge.l[l];
            gen_layer_lds(p.blob, L, 0, L.in, cur, nxt, TS, t, [&](int o) { return p.blob[L.boff + o]; });
            float* tmp = cur;
            cur = nxt, nxt = tmp;
        }
        // cur = e' [EF][T]
        if (live)
            for (int q = 0; q < p.EF; ++q) p.e_new[(size_t)k * p.e_new_ld + q] = cur[q * TS + t];
        const float* e_lds = cur;   // e' [EF][T]: read by the node MLP's first layer and by the classifier before anything overwrites it
        // ---- node MLP, first layer (models/mpn.py:97-98): Q[row] + W_e e' -> the free buffer ---------------------------------------
        if (p.h_new)
            gen_layer_lds(p.blob, p.node.l[0], p.k0_node, p.EF, e_lds, nxt, TS, t,
                          [&](int o) { return o < p.o1n ? tab[2 * p.o1e + o] : 0.f; });
        // ---- classifier on e' (models/mpn.py:290-293): widths <= 16 (gen_fused_ok), the thread's activations in registers ------------
        if (p.logits) {
            float va[16], vb[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) va[q] = q < p.EF ? e_lds[q * TS + t] : 0.f;
            int wv = p.EF;
            for (int l = 0; l < p.cls.n; ++l) {
                const GenLayerDesc& L = p.cls.l[l];
                const float* __restrict__ Wt = p.blob + L.woff;
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    float acc = 0.f;
                    if (o < L.out) {
                        acc = p.blob[L.boff + o];
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (q < wv) acc = fmaf(Wt[(size_t)q * L.op + o], va[q], acc);
                        if (L.relu) acc = fmaxf(acc, 0.f);
                    }
                    vb[o] = acc;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) va[q] = vb[q];
                wv = L.out;
            }
            if (live) p.logits[k] = va[0];
        }
        // ---- node MLP, remaining layers; then the tile's contribution to h' ------------------------------------------------------
        if (p.h_new) {
            float* a = nxt;     // the first node layer's output
            float* b = cur;     // e' is dead now
            for (int l = 1; l < p.node.n; ++l) {
                const GenLayerDesc& L = p.node.l[l];
                gen_layer_lds(p.blob, L, 0, L.in, a, b, TS, t, [&](int o) { return p.blob[L.boff + o]; });
                float* tmp = a;
                a = b, b = tmp;
            }
            if (!live)
                for (int c = 0; c < p.H; ++c) a[c * TS + t] = ident;    // a dead thread contributes the identity
            __syncthreads();
            if (reducer) {
                const int lo = rp * chunk, hi = min(lo + chunk, T);
                const float* row = a + rc * TS;
                if (p.agg == GNNCCA_AGG_MAX)
                    for (int u = lo; u < hi; ++u) hacc = fmaxf(hacc, row[u]);
                else
                    for (int u = lo; u < hi; ++u) hacc += row[u];
            }
        }
        __syncthreads();   // the activation buffers are rewritten by the next tile
    }
    if (!p.h_new) return;
    // ---- combine the parts in fixed order, finish the aggregator, store h', project for the next step ----------------------------------
    float* s_hin = s_x;
    float* s_red = s_x + p.hin_w;
    if (reducer) s_red[rp * p.H + rc] = hacc;
    __syncthreads();
    const int deg = seg_t - seg_s;
    const int hoff = p.h0 ? p.H : 0;
    if (t < p.H) {
        float v = s_red[t];
        for (int u = 1; u < parts; ++u) v = p.agg == GNNCCA_AGG_MAX ? fmaxf(v, s_red[u * p.H + t]) : v + s_red[u * p.H + t];
        if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
        if (deg == 0) v = 0.f;
        p.h_new[(size_t)node * p.H + t] = v;
        s_hin[hoff + t] = v;
        if (p.h0) s_hin[t] = p.h0[(size_t)node * p.H + t];
    }
    __syncthreads();
    if (p.tab_out) gen_project_node(p, s_hin, node);
}

// can the fused step run this configuration?  (every width it keeps per thread in LDS within the budget; at least one layer in
// the two message-passing MLPs; H <= T so that every channel has a reducer thread)
static bool gen_fused_ok(const gnncca_mpn_dims* d, int* T_out, int* wmax_out) {
    if (d->edge_mlp.n_layers < 1 || d->node_mlp.n_layers < 1 || d->num_enc_steps < 1) return false;
    int wmax = std::max(std::max((d->reattach_edges ? 2 : 1) * d->edge_dim, d->node_dim), 4);
    for (int l = 0; l < d->edge_mlp.n_layers; ++l) wmax = std::max(wmax, (int)d->edge_mlp.layers[l].out_dim);
    for (int l = 0; l < d->node_mlp.n_layers; ++l) wmax = std::max(wmax, (int)d->node_mlp.layers[l].out_dim);
    for (int l = 0; l < d->cls_edge.n_layers; ++l) wmax = std::max(wmax, (int)d->cls_edge.layers[l].out_dim);
    if (wmax > 128 || d->edge_dim > 16) return false;
    for (int l = 0; l < d->cls_edge.n_layers; ++l)
        if (d->cls_edge.layers[l].out_dim > 16) return false;   // the classifier's hidden layers live in 16 registers per thread
    if (d->edge_mlp.layers[d->edge_mlp.n_layers - 1].out_dim != d->edge_dim || d->node_mlp.layers[d->node_mlp.n_layers - 1].out_dim != d->node_dim)
        return false;
    const int T = wmax <= 32 ? 256 : 128;
    if (d->node_dim > T) return false;
    *T_out = T;
    *wmax_out = wmax;
    return true;
}

static void gen_fill_mlp(GenMlpDesc* m, const gnncca_mlp& mlp, const int32_t* woff, const int32_t* boff) {
    m->n = mlp.n_layers;
    for (int l = 0; l < mlp.n_layers; ++l) {
        m->l[l].woff = woff[l];
        m->l[l].boff = boff[l];
        m->l[l].in = mlp.layers[l].in_dim;
        m->l[l].out = mlp.layers[l].out_dim;
        m->l[l].op = (mlp.layers[l].out_dim + 7) / 8 * 8;
        m->l[l].relu = mlp.layers[l].relu;
    }
}

}  // namespace gnncca
