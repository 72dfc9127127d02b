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
    int k0, kn;                          // the rows of Wt this kernel uses per edge (first layers of the edge / node MLP: the e block only)
    int lw, lb;                          // where the workgroup stages them in LDS (float offsets into s_w): [kn][op] weights, [op] bias
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
    GenMlpDesc edge, node, cls, enc;   // enc: encoder.edge_mlp (models/mpn.py:137), run by step 1 on the raw edge attributes when `edge_attr` is set
    const float* edge_attr;   // [E][edge_in] or null: the edge input comes from e_a / e_b
    int edge_in;
    float* e0_out;            // [E][EF] encoded edge features to HBM (reattach_initial_edges, debug trace), or null
    int k0_edge, k0_node;   // first column of the e block in the first layer's weight (2 * hin_w, hin_w)
    float* e_new;           // [E][e_new_ld] latent edge features after this step (caller's edge order)
    int e_new_ld;
    float* logits;          // [E] or null: this step does not classify
    float* h_new;           // [N][H] or null: the node update is not needed (last step)
    const float* h0;        // [N][H]: initial node features (reattach_nodes), else null
    int H, EF, hin_w, agg;
    int lds_stride;         // T + 1
    int wmax;               // widest activation vector a thread keeps in LDS (gen_fused_ok)
    int w_floats;           // LDS floats of the staged weights and biases (all layers of the three MLPs), padded to 4
    int w_used, step_w;     // ... unpadded; float offset of the stage image in the blob (GenBlobHeader::step_w)
};

// one dense layer on the tile: s_out[o][t] = act(init(o) + sum_k W[k][o] * s_in[k][t]), four outputs per pass; the layer's weights were
// staged in LDS by the workgroup (s_w + L.lw: [kn][op]) and are read as 16-byte broadcasts -- straight from the blob every pass was a
// dependent L2 round trip per (k, four outputs): 38 us per step at node latent 64 where this form takes a few
template <typename Init>
__device__ __forceinline__ void gen_layer_lds(const float* s_w, const GenLayerDesc& L, const float* s_in, float* s_out, int TS, int t, Init init) {
    const float* Wl = s_w + L.lw;
    const int K = L.kn, op = L.op, O = L.out;
    const bool relu = L.relu != 0;
    if (K <= 16) {
        // short inputs (the e block of a first layer: <= 16 columns): the thread's inputs in registers, EIGHT outputs per pass --
        // per pass K x (two 16-byte weight broadcasts + 8 FMAs) and no activation reads at all
        float xr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = q < K ? s_in[q * TS + t] : 0.f;
        for (int o0 = 0; o0 < O; o0 += 8) {   // op is a multiple of 8: the padded weight columns are zeros
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = init(o0 + j);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wl + k * op + o0);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wl + k * op + o0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(w0[j], xr[k], acc[j]), acc[4 + j] = fmaf(w1[j], xr[k], acc[4 + j]);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (o0 + j < O) s_out[(o0 + j) * TS + t] = relu ? fmaxf(acc[j], 0.f) : acc[j];
        }
        return;
    }
    for (int o0 = 0; o0 < O; o0 += 4) {
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = init(o0 + j);
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float x = s_in[k * TS + t];
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wl + k * op + o0);   // wave-uniform address: a broadcast; op is a multiple of 8
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[j], x, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (o0 + j < O) s_out[(o0 + j) * TS + t] = relu ? fmaxf(acc[j], 0.f) : acc[j];
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
        // (the weight loads in flight in batches of 32: one load per iteration behind the FMA chain was a dependent L2 round trip per k --
        // 64 of them at node latent 64, most of a step launch's time)
        int k = 0;
        for (; k + 32 <= p.hin_w; k += 32) {
            float w[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = Wt[(size_t)(k + u) * op + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = fmaf(w[u], s_hin[k + u], acc);
        }
        for (; k + 8 <= p.hin_w; k += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = Wt[(size_t)(k + u) * op + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(w[u], s_hin[k + u], acc);
        }
        for (; k < p.hin_w; ++k) acc = fmaf(Wt[(size_t)k * op + o], s_hin[k], acc);
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

// The node side of the encoder behind the first-layer GEMM, the step-1 projection tables and the plan's flag fold in ONE launch
// (rounds 1-3: reduce_bias_act_kernel, one gen_dense_kernel per further layer, gen_plan_finish_kernel; round 4's first fused form added
// gen_project_kernel): workgroup = one node -- split-K slab sum + bias + ReLU (or the raw feature row when the first layer did not go
// through the GEMM), the remaining encoder layers with a thread per output, h0 to HBM, then gen_project_node; the LAST workgroup of the
// launch folds the plan's per-block findings and repairs an unsorted plan (plan_finish), as the MFMA family's tails do.
struct GenTailParams {
    const float* x;          // [N][node_in]
    const float* part;       // [ks][N][O0] split-K partials of the first layer, or null: start from x with layer 0
    int ks, first_layer;     // first_layer: index of the first encoder layer this kernel computes (1 behind the GEMM, else 0)
    GenMlpDesc enc;          // encoder.node_mlp, all layers
    int node_in, wmax_enc;   // widest vector of the chain (LDS ping-pong buffers)
    float* h0;               // [N][H]
    float* trace_h;          // or null
    const long long* ei;     // plan_finish
    int* seg_ptr;
    int* col32;
    int* perm;
    int* cursor;
    unsigned* flags;
    const unsigned* blockflags;
    int E;
};
__global__ __launch_bounds__(256) void gen_node_tail_kernel(const GenStepParams p, const GenTailParams q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x;
    if ((int)blockIdx.x == p.N) {   // the plan workgroup
        plan_finish(q.ei, q.E, p.N, q.seg_ptr, q.col32, q.perm, q.cursor, q.flags, q.blockflags, reinterpret_cast<unsigned*>(smem));
        return;
    }
    const int node = blockIdx.x;
    float* a = smem;
    float* b = smem + q.wmax_enc;
    int w;
    if (q.part) {   // slab sum in slab order + bias + activation of the GEMM's layer
        const GenLayerDesc& L0 = q.enc.l[0];
        w = L0.out;
        for (int o = t; o < w; o += 256) {
            float v = p.blob[L0.boff + o];
            int s = 0;
            for (; s + 8 <= q.ks; s += 8) {   // eight slabs in flight, summed in slab order
                float pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) pv[u] = q.part[((size_t)(s + u) * p.N + node) * w + o];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += pv[u];
            }
            for (; s < q.ks; ++s) v += q.part[((size_t)s * p.N + node) * w + o];
            a[o] = L0.relu ? fmaxf(v, 0.f) : v;
        }
    } else {
        w = q.node_in;
        for (int k = t; k < w; k += 256) a[k] = q.x[(size_t)node * w + k];
    }
    __syncthreads();
    for (int l = q.first_layer; l < q.enc.n; ++l) {
        const GenLayerDesc& L = q.enc.l[l];
        const float* __restrict__ Wt = p.blob + L.woff;
        for (int o = t; o < L.out; o += 256) {
            float acc = p.blob[L.boff + o];
            int k = 0;
            for (; k + 16 <= w; k += 16) {   // sixteen weight loads in flight per batch (see gen_project_node)
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = Wt[(size_t)(k + u) * L.op + o];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fmaf(wv[u], a[k + u], acc);
            }
            for (; k < w; ++k) acc = fmaf(Wt[(size_t)k * L.op + o], a[k], acc);
            b[o] = L.relu ? fmaxf(acc, 0.f) : acc;
        }
        __syncthreads();
        float* tmp = a;
        a = b, b = tmp;
        w = L.out;
    }
    // a = the encoder's node output [H] (w == H)
    for (int c = t; c < p.H; c += 256) {
        q.h0[(size_t)node * p.H + c] = a[c];
        if (q.trace_h) q.trace_h[(size_t)node * p.H + c] = a[c];
    }
    float* s_hin = b;   // (wmax_enc >= hin_w: the host sizes it so)
    for (int k = t; k < p.hin_w; k += 256) s_hin[k] = a[k < p.H ? k : k - p.H];   // before step 1 initial == latent (mpn.py:271-285)
    __syncthreads();
    gen_project_node(p, s_hin, node);
}

__global__ __launch_bounds__(256) void gen_step_fused_kernel(const GenStepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = blockDim.x, TS = p.lds_stride, t = threadIdx.x;
    const int node = blockIdx.x;
    // LDS: two activation buffers [wmax][TS], then [hin_w] + [parts][H] scratch of the epilogue
    const int wmax = p.wmax;
    float* s_a = smem;
    float* s_b = s_a + (size_t)wmax * TS;
    float* s_w = s_b + (size_t)wmax * TS;       // staged weights and biases of every layer used per edge
    float* s_tab = s_w + p.w_floats;            // this node's row of the projection table: P_src | P_dst | Q
    float* s_x = s_tab + (p.tab_ld + 3) / 4 * 4;   // [hin_w] then [parts][H]
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
    {   // stage the weights and this node's table row: ONE coalesced copy of the blob's stage image (pack.cpp: gen_step_block keeps it in the order
        // and shape gen_fill_mlp assigns the LDS offsets in), every load in flight before the first store
        const float* __restrict__ src = p.blob + p.step_w;
        for (int i = t; i < p.w_used; i += T) s_w[i] = src[i];
        for (int i = t; i < p.tab_ld; i += T) s_tab[i] = p.tab_in[(size_t)node * p.tab_ld + i];
        __syncthreads();
    }
    const float* tab = s_tab;   // this node's P_src | . | Q, staged below (read once per output pass: from HBM / L2 that was a dependent
                                // round trip per pass -- sixteen per tile at node latent 64)
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
        float* cur = s_a;
        float* nxt = s_b;
        if (p.edge_attr) {
            // ---- step 1: the edge encoder on the raw attributes (models/mpn.py:137), then cat(e0, e0) when reattaching (initial == latent) ----
            for (int q = 0; q < p.edge_in; ++q) cur[q * TS + t] = p.edge_attr[(size_t)k * p.edge_in + q];
            for (int l = 0; l < p.enc.n; ++l) {
                const GenLayerDesc& L = p.enc.l[l];
                gen_layer_lds(s_w, L, cur, nxt, TS, t, [&](int o) { return s_w[L.lb + min(o, L.op - 1)]; });
                float* tmp = cur;
                cur = nxt, nxt = tmp;
            }
            if (p.e0_out && live)
                for (int q = 0; q < p.EF; ++q) p.e0_out[(size_t)k * p.EF + q] = cur[q * TS + t];
            if (p.e_b_w > 0)
                for (int q = 0; q < p.EF; ++q) cur[(p.EF + q) * TS + t] = cur[q * TS + t];
        } else {
            // ---- edge MLP (models/mpn.py:68-69): inputs cat(e0, e) or e, k-major ---------------------------------------------------
            for (int q = 0; q < p.e_a_w; ++q) cur[q * TS + t] = p.e_a[(size_t)k * p.e_a_ld + q];
            for (int q = 0; q < p.e_b_w; ++q) cur[(p.e_a_w + q) * TS + t] = p.e_b[(size_t)k * p.e_b_ld + q];
        }
        const float* __restrict__ pdst = p.tab_in + (size_t)j * p.tab_ld + p.o1e;
        // the gathered P_dst[col] row parks in the OUTPUT buffer's own slots (all loads in flight at once); every output pass reads its
        // four slots before it overwrites them
        for (int o = 0; o < p.o1e; ++o) nxt[o * TS + t] = pdst[o];
        gen_layer_lds(s_w, p.edge.l[0], cur, nxt, TS, t, [&](int o) { return o < p.o1e ? tab[o] + nxt[o * TS + t] : 0.f; });
        {
            float* tmp = cur;
            cur = nxt, nxt = tmp;
        }
        for (int l = 1; l < p.edge.n; ++l) {
            const GenLayerDesc& L = p.edge.l[l];
            gen_layer_lds(s_w, L, cur, nxt, TS, t, [&](int o) { return s_w[L.lb + min(o, L.op - 1)]; });
            float* tmp = cur;
            cur = nxt, nxt = tmp;
        }
        // cur = e' [EF][T]
        if (live)
            for (int q = 0; q < p.EF; ++q) p.e_new[(size_t)k * p.e_new_ld + q] = cur[q * TS + t];
        const float* e_lds = cur;   // e' [EF][T]: read by the node MLP's first layer and by the classifier before anything overwrites it
        // ---- node MLP, first layer (models/mpn.py:97-98): Q[row] + W_e e' -> the free buffer ---------------------------------------
        if (p.h_new)
            gen_layer_lds(s_w, p.node.l[0], e_lds, nxt, TS, t, [&](int o) { return o < p.o1n ? tab[2 * p.o1e + o] : 0.f; });
        // ---- classifier on e' (models/mpn.py:290-293): widths <= 16 (gen_fused_ok), the thread's activations in registers ------------
        if (p.logits) {
            float va[16], vb[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) va[q] = q < p.EF ? e_lds[q * TS + t] : 0.f;
            int wv = p.EF;
            for (int l = 0; l < p.cls.n; ++l) {
                const GenLayerDesc& L = p.cls.l[l];
                const float* Wl = s_w + L.lw;
                const int op = L.op;
                const bool relu = L.relu != 0;
#pragma unroll
                for (int o0 = 0; o0 < 16; o0 += 8) {   // eight outputs per pass, weights as 16-byte broadcasts (op is a multiple of 8)
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                    if (o0 < L.out) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = s_w[L.lb + o0 + j];
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (q < wv) {
                                const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wl + q * op + o0);
                                const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wl + q * op + o0 + 4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w0[j], va[q], acc[j]), acc[4 + j] = fmaf(w1[j], va[q], acc[4 + j]);
                            }
                        if (relu) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) vb[o0 + j] = acc[j];   // (columns beyond L.out: padded weights and bias are zero)
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
                gen_layer_lds(s_w, L, a, b, TS, t, [&](int o) { return s_w[L.lb + min(o, L.op - 1)]; });
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

// dynamic LDS of one gen_step_fused_kernel workgroup (floats -> bytes): the one formula gen_fused_ok admits by and the launch sizes by
constexpr size_t kGenFusedLdsMax = 160 * 1024;
static inline size_t gen_fused_lds_bytes(int wmax_padded, int T, int w_floats, int tab_ld, int hin_w, int H) {
    const int parts = std::max(1, T / std::max(H, 1));
    return ((size_t)2 * wmax_padded * (T + 1) + (size_t)w_floats + (size_t)((tab_ld + 3) / 4 * 4) + (size_t)hin_w + (size_t)parts * H) * sizeof(float);
}

// can the fused step run this configuration?  (every width it keeps per thread in LDS within the budget; at least one layer in
// the two message-passing MLPs; H <= T so that every channel has a reducer thread)
static bool gen_fused_ok(const gnncca_mpn_dims* d, int64_t n_nodes, int* T_out, int* wmax_out) {
    if (d->edge_mlp.n_layers < 1 || d->node_mlp.n_layers < 1 || d->num_enc_steps < 1) return false;
    int wmax = std::max(std::max((d->reattach_edges ? 2 : 1) * d->edge_dim, d->node_dim), 4);
    for (int l = 0; l < d->edge_mlp.n_layers; ++l) wmax = std::max(wmax, (int)d->edge_mlp.layers[l].out_dim);
    for (int l = 0; l < d->node_mlp.n_layers; ++l) wmax = std::max(wmax, (int)d->node_mlp.layers[l].out_dim);
    for (int l = 0; l < d->cls_edge.n_layers; ++l) wmax = std::max(wmax, (int)d->cls_edge.layers[l].out_dim);
    wmax = std::max(wmax, (int)d->edge_in);
    for (int l = 0; l < d->enc_edge.n_layers; ++l) wmax = std::max(wmax, (int)d->enc_edge.layers[l].out_dim);
    if (wmax > 128 || d->edge_dim > 16) return false;
    if (d->node_in > 8192) return false;   // (the encoder tail keeps a node's widest vector twice in LDS)
    for (int l = 0; l < d->enc_node.n_layers; ++l)
        if (d->enc_node.layers[l].out_dim > 8192) return false;
    for (int l = 0; l < d->cls_edge.n_layers; ++l)
        if (d->cls_edge.layers[l].out_dim > 16) return false;   // the classifier's hidden layers live in 16 registers per thread
    if (d->edge_mlp.layers[d->edge_mlp.n_layers - 1].out_dim != d->edge_dim || d->node_mlp.layers[d->node_mlp.n_layers - 1].out_dim != d->node_dim)
        return false;
    // edges per tile = threads per workgroup.  A single frame-sized graph has one workgroup per CU at most: everything is exposed
    // latency there and the widest tile (one pass over a dense-256 node's 255 edges) is the fastest -- 132 KB of LDS at width 64;
    // batches want two workgroups per CU instead (66 KB)
    int T = (wmax <= 32 || (wmax <= 64 && n_nodes <= 1024)) ? 256 : 128;
    size_t wf = 0;
    {   // the weights every edge uses must fit their LDS stage (40 KB) next to the activations
        const int ef_in = (d->reattach_edges ? 2 : 1) * d->edge_dim;
        auto add = [&](const gnncca_mlp& m, int kn_first) {
            for (int l = 0; l < m.n_layers; ++l)
                wf += (size_t)((l == 0 && kn_first >= 0 ? kn_first : m.layers[l].in_dim) + 1) * (size_t)((m.layers[l].out_dim + 7) / 8 * 8);
        };
        add(d->edge_mlp, ef_in), add(d->node_mlp, d->edge_dim), add(d->cls_edge, -1), add(d->enc_edge, -1);
        if (wf > 10240) return false;
    }
    const int wpad = (wmax + 3) / 4 * 4;   // keeps the LDS regions behind the activation buffers 16-byte aligned
    // The WHOLE LDS footprint of gen_step_fused_kernel, as the launch code sizes it (gen_fused_lds_bytes below is that formula): two
    // k-major activation buffers, the weight stage, the node's table row, its hin vector and the reduction parts.  It must fit the 160 KB
    // a CU has: the widest tile first, then the 128-edge tile, else the op-by-op path (a launch that asks for more would fail outright).
    const int tab_ld = 2 * d->edge_mlp.layers[0].out_dim + d->node_mlp.layers[0].out_dim;
    const int hin_w = (d->reattach_nodes ? 2 : 1) * d->node_dim;
    for (;; T = 128) {
        if (d->node_dim <= T && gen_fused_lds_bytes(wpad, T, (int)((wf + 3) / 4 * 4), tab_ld, hin_w, d->node_dim) <= kGenFusedLdsMax) break;
        if (T == 128) return false;
    }
    *T_out = T;
    *wmax_out = wpad;
    return true;
}

// descriptors of one MLP; `k0_first` / `kn_first`: the rows of the FIRST layer's weight used per edge (-1: all of them)
static void gen_fill_mlp(GenMlpDesc* m, const gnncca_mlp& mlp, const int32_t* woff, const int32_t* boff, int k0_first, int kn_first, int* lds_floats) {
    m->n = mlp.n_layers;
    for (int l = 0; l < mlp.n_layers; ++l) {
        GenLayerDesc& L = m->l[l];
        L.woff = woff[l];
        L.boff = boff[l];
        L.in = mlp.layers[l].in_dim;
        L.out = mlp.layers[l].out_dim;
        L.op = (mlp.layers[l].out_dim + 7) / 8 * 8;
        L.relu = mlp.layers[l].relu;
        L.k0 = (l == 0 && k0_first >= 0) ? k0_first : 0;
        L.kn = (l == 0 && k0_first >= 0) ? kn_first : L.in;
        L.lw = *lds_floats;
        *lds_floats += L.kn * L.op;
        L.lb = *lds_floats;
        *lds_floats += L.op;
    }
}

}  // namespace gnncca
