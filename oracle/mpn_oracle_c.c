/* CPU ORACLE, C flavour, for the GNN-CCA message-passing hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library (through oracle/c_oracle.py),
 * and only as a checker / the timed CPU baseline.  Parity status: PINNED -- tests/test_oracle_golden.py holds it to the golden
 * vectors the reference's own MOTMPNet produced (the .npz files under tests/golden) for every case it supports.
 *
 * What it is: the same eval-mode forward as oracle/mpn_oracle.py (reference lines below), but in the FUSED, split-weight
 * form a hand-written CPU implementation would take -- the [E,70] / [E,38] concatenations of models/mpn.py:68,97 are never
 * materialised -- with OpenMP over nodes and edges.  It is SURVEY.md 8(d)'s "flavour (ii)" CPU baseline: how fast the path
 * can go on the host's cores when it is written for them, next to flavour (i) (oracle.TorchOracle, the reference's own torch
 * ops).  fp32 arithmetic, sequential sums in edge order per segment (the order torch's CPU index_add_ sums in).
 *
 * Reference lines followed (relative to /root/reference):
 *   models/mlp.py:4-28      Linear [+ BatchNorm1d eval] [+ ReLU] stacks                        -> mlp_rows()
 *   models/mpn.py:128-142   encoder / classifier (MLPGraphIndependent)                         -> mlp_rows() on nodes / edges
 *   models/mpn.py:59-69     EdgeModel: Linear(cat[x[row], x[col], e]) + ReLU                   -> split: P_src[row] + P_dst[col] + W_ee e
 *   models/mpn.py:71-101    NodeModel: Linear(cat[x[row], e']) + ReLU, aggregate by `row`      -> split: Q[row] + W_ne e'
 *   models/mpn.py:192-202   sum / mean / max aggregators, empty segment -> 0
 *   models/mpn.py:250-299   forward: encode, L steps (reattach: initial first), classify from step L - n_cls + 1, L == 0
 * Scope: single-layer edge / node MLPs inside the MPN (every configuration the reference ships); deeper ones return -2.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int in_dim, out_dim, has_bn, relu;
    const float *W, *b;                      /* [out][in], [out] */
    const float *gamma, *beta, *mean, *var;  /* BatchNorm1d (eval), or NULL */
} oc_layer;

typedef struct {
    int n_layers;
    oc_layer layers[8];
} oc_mlp;

typedef struct {
    oc_mlp enc_node, enc_edge, edge_mlp, node_mlp, cls_edge;
    int agg;       /* 0 sum, 1 mean, 2 max */
    int L, n_cls, reattach_nodes, reattach_edges;
} oc_model;

#define BN_EPS 1e-5f

/* y[rows][out] = MLP(x[rows][in]); tmp buffers allocated inside */
static float* mlp_rows(const oc_mlp* m, const float* x, int64_t rows, int in_dim, int* out_dim) {
    const float* cur = x;
    float* owned = NULL;
    int cur_dim = in_dim;
    for (int l = 0; l < m->n_layers; ++l) {
        const oc_layer* L = &m->layers[l];
        float* y = (float*)malloc(sizeof(float) * (size_t)(rows > 0 ? rows : 1) * L->out_dim);
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            const float* xr = cur + (size_t)r * cur_dim;
            float* yr = y + (size_t)r * L->out_dim;
            for (int o = 0; o < L->out_dim; ++o) {
                const float* w = L->W + (size_t)o * L->in_dim;
                float s = 0.f;
                for (int k = 0; k < L->in_dim; ++k) s += xr[k] * w[k];
                s += L->b[o];
                if (L->has_bn) s = (s - L->mean[o]) / sqrtf(L->var[o] + BN_EPS) * L->gamma[o] + L->beta[o];
                if (L->relu && s < 0.f) s = 0.f;
                yr[o] = s;
            }
        }
        free(owned);
        owned = y;
        cur = y;
        cur_dim = L->out_dim;
    }
    if (m->n_layers == 0) {  /* absent MLP: identity (models/mpn.py:133-140 passes the input through) */
        owned = (float*)malloc(sizeof(float) * (size_t)(rows > 0 ? rows : 1) * in_dim);
        memcpy(owned, x, sizeof(float) * (size_t)rows * in_dim);
    }
    *out_dim = cur_dim;
    return owned;
}

/* logits_out: [n_out][E]; returns the number of classified steps written, or < 0 */
int oc_forward(const oc_model* md, const float* x, const int64_t* edge_index, const float* edge_attr, int64_t N, int64_t E,
               int node_in, int edge_in, float* logits_out) {
    if (md->edge_mlp.n_layers != 1 || md->node_mlp.n_layers != 1) return -2;
    if (md->edge_mlp.layers[0].has_bn || md->node_mlp.layers[0].has_bn) return -2;
    const int64_t* row = edge_index;
    const int64_t* col = edge_index + E;
    for (int64_t k = 0; k < E; ++k)
        if (row[k] < 0 || row[k] >= N || col[k] < 0 || col[k] >= N) return -1;
    int H, F;
    float* h = mlp_rows(&md->enc_node, x, N, node_in, &H);          /* mpn.py:270 */
    float* e = mlp_rows(&md->enc_edge, edge_attr, E, edge_in, &F);
    float* h0 = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * H);
    float* e0 = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * F);
    memcpy(h0, h, sizeof(float) * (size_t)N * H);
    memcpy(e0, e, sizeof(float) * (size_t)E * F);
    const int nf = md->reattach_nodes ? 2 : 1, ef = md->reattach_edges ? 2 : 1;
    const int HI = nf * H, EI = ef * F;
    const oc_layer* Le = &md->edge_mlp.layers[0];   /* [F'][2 HI + EI] */
    const oc_layer* Ln = &md->node_mlp.layers[0];   /* [H'][HI + F'] */
    if (Le->in_dim != 2 * HI + EI || Ln->in_dim != HI + Le->out_dim || Ln->out_dim != H || Le->out_dim != F) {
        free(h), free(e), free(h0), free(e0);
        return -3;
    }
    /* CSR by source row, stable (edge order inside a segment = the caller's order) */
    int64_t* seg = (int64_t*)calloc((size_t)N + 2, sizeof(int64_t));
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(E > 0 ? E : 1));
    for (int64_t k = 0; k < E; ++k) seg[row[k] + 1]++;
    for (int64_t i = 0; i < N; ++i) seg[i + 1] += seg[i];
    {
        int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
        memcpy(cur, seg, sizeof(int64_t) * (size_t)N);
        for (int64_t k = 0; k < E; ++k) order[cur[row[k]]++] = k;
        free(cur);
    }
    float* Ps = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * F);
    float* Pd = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * F);
    float* Q = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * H);
    float* hn = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * H);
    float* en = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * F);
    const int first = md->L - md->n_cls + 1;   /* mpn.py:277 */
    int n_out = 0, cls_dim = 0;
    for (int step = 1; step <= md->L; ++step) {
        /* per-node projections of cat(h0, h) (reattach: initial first, mpn.py:285) */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            float hin[128];
            for (int c = 0; c < H; ++c) {
                if (md->reattach_nodes) hin[c] = h0[(size_t)i * H + c], hin[H + c] = h[(size_t)i * H + c];
                else hin[c] = h[(size_t)i * H + c];
            }
            for (int f = 0; f < F; ++f) {
                const float* w = Le->W + (size_t)f * Le->in_dim;
                float s = 0.f, t = 0.f;
                for (int c = 0; c < HI; ++c) s += hin[c] * w[c], t += hin[c] * w[HI + c];
                Ps[(size_t)i * F + f] = s + Le->b[f];
                Pd[(size_t)i * F + f] = t;
            }
            for (int o = 0; o < H; ++o) {
                const float* w = Ln->W + (size_t)o * Ln->in_dim;
                float s = 0.f;
                for (int c = 0; c < HI; ++c) s += hin[c] * w[c];
                Q[(size_t)i * H + o] = s + Ln->b[o];
            }
        }
        /* one pass over a node's segment: edge update, message, aggregation */
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t i = 0; i < N; ++i) {
            float acc[128];
            const int64_t s0 = seg[i], s1 = seg[i + 1];
            for (int o = 0; o < H; ++o) acc[o] = md->agg == 2 ? -INFINITY : 0.f;
            for (int64_t q = s0; q < s1; ++q) {
                const int64_t k = order[q];
                const int64_t j = col[k];
                float ein[32], ev[16];
                for (int f = 0; f < F; ++f) {
                    if (md->reattach_edges) ein[f] = e0[(size_t)k * F + f], ein[F + f] = e[(size_t)k * F + f];  /* mpn.py:283 */
                    else ein[f] = e[(size_t)k * F + f];
                }
                for (int f = 0; f < F; ++f) {
                    const float* w = Le->W + (size_t)f * Le->in_dim + 2 * HI;
                    float s = Ps[(size_t)i * F + f] + Pd[(size_t)j * F + f];
                    for (int g = 0; g < EI; ++g) s += w[g] * ein[g];
                    ev[f] = (Le->relu && s < 0.f) ? 0.f : s;
                    en[(size_t)k * F + f] = ev[f];
                }
                for (int o = 0; o < H; ++o) {
                    const float* w = Ln->W + (size_t)o * Ln->in_dim + HI;
                    float s = Q[(size_t)i * H + o];
                    for (int f = 0; f < F; ++f) s += w[f] * ev[f];
                    if (Ln->relu && s < 0.f) s = 0.f;
                    if (md->agg == 2) acc[o] = s > acc[o] ? s : acc[o];
                    else acc[o] += s;
                }
            }
            for (int o = 0; o < H; ++o) {
                float v = acc[o];
                if (s1 == s0) v = 0.f;                                   /* rows that receive nothing are 0 */
                else if (md->agg == 1) v = v / (float)(s1 - s0);        /* scatter_mean */
                hn[(size_t)i * H + o] = v;
            }
        }
        { float* t = h; h = hn; hn = t; }
        { float* t = e; e = en; en = t; }
        if (step >= first) {   /* mpn.py:290-293 */
            float* lg = mlp_rows(&md->cls_edge, e, E, F, &cls_dim);
            if (cls_dim != 1) { free(lg); n_out = -4; break; }
            memcpy(logits_out + (size_t)n_out * E, lg, sizeof(float) * (size_t)E);
            free(lg);
            ++n_out;
        }
    }
    if (md->L == 0) {   /* mpn.py:295-297 */
        float* lg = mlp_rows(&md->cls_edge, e, E, F, &cls_dim);
        if (cls_dim == 1) memcpy(logits_out, lg, sizeof(float) * (size_t)E), n_out = 1;
        else n_out = -4;
        free(lg);
    }
    free(h), free(e), free(h0), free(e0), free(seg), free(order), free(Ps), free(Pd), free(Q), free(hn), free(en);
    return n_out;
}
