// pack.cpp -- host side of libgnncca_mpn: dims validation, weight packing, workspace carve-up.
// Pure C++ (no HIP calls): testable on a machine without a GPU.
//
// What the packing does, with the reference lines it derives from:
//   * eval-mode BatchNorm1d (models/mlp.py:14-15) is folded into the preceding Linear:
//       y = ((W x + b) - mean) / sqrt(var + 1e-5) * gamma + beta  ==  (s.W) x + ((b - mean) s + beta),
//       s = gamma / sqrt(var + 1e-5)                       (computed in double, rounded once to fp32)
//   * the edge MLP weight [EF][nf*2H + ef*EF] is split by the cat order of models/mpn.py:68
//       [ x[row] | x[col] | edge_attr ]  ->  W_src, W_dst (per-node projections) and W_ee (per edge)
//   * the node MLP weight [H][nf*H + EF] is split by the cat order of models/mpn.py:97
//       [ x[row] | edge_attr ]           ->  W_nx (per-node projection Q) and W_ne (per edge, MFMA B operand)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "internal.h"

namespace gnncca {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int mlp_param_count(const gnncca_mlp& m) {
    int n = 0;
    for (int i = 0; i < m.n_layers; ++i) n += 2 + (m.layers[i].has_bn ? 4 : 0);
    return n;
}

static bool mlp_chain_ok(const gnncca_mlp& m, int in_dim, int out_dim) {
    if (m.n_layers < 0 || m.n_layers > GNNCCA_MAX_LAYERS) return false;
    if (m.n_layers == 0) return in_dim == out_dim;
    int cur = in_dim;
    for (int i = 0; i < m.n_layers; ++i) {
        const gnncca_layer& l = m.layers[i];
        if (l.in_dim != cur || l.out_dim <= 0) return false;
        if ((l.relu != 0) != (l.out_dim != 1)) return false;  // models/mlp.py:17
        if (l.has_bn && l.out_dim == 1) return false;         // models/mlp.py:14
        cur = l.out_dim;
    }
    return cur == out_dim;
}

bool dims_valid(const gnncca_mpn_dims* d) {
    if (!d || d->abi_version != GNNCCA_ABI_VERSION) return false;
    if (d->node_in <= 0 || d->edge_in <= 0 || d->node_dim <= 0 || d->edge_dim <= 0) return false;
    if (d->agg < GNNCCA_AGG_SUM || d->agg > GNNCCA_AGG_MAX) return false;
    if (d->num_enc_steps < 0) return false;
    const int nf = d->reattach_nodes ? 2 : 1, ef = d->reattach_edges ? 2 : 1;
    if (!mlp_chain_ok(d->enc_node, d->node_in, d->node_dim)) return false;
    if (!mlp_chain_ok(d->enc_edge, d->edge_in, d->edge_dim)) return false;
    if (d->edge_mlp.n_layers < 1 || d->node_mlp.n_layers < 1 || d->cls_edge.n_layers < 1) return false;
    if (!mlp_chain_ok(d->edge_mlp, nf * 2 * d->node_dim + ef * d->edge_dim, d->edge_dim)) return false;  // mpn.py:213
    if (!mlp_chain_ok(d->node_mlp, nf * d->node_dim + d->edge_dim, d->node_dim)) return false;           // mpn.py:215
    if (!mlp_chain_ok(d->cls_edge, d->edge_dim, 1)) return false;
    return true;
}

Family classify(const gnncca_mpn_dims* d) {
    if (!dims_valid(d)) return kFamilyNone;
    if (d->node_dim != kH || d->edge_dim != kEF) return kFamilyGeneric;
    if (d->edge_in > kMaxEdgeIn) return kFamilyGeneric;
    if (d->enc_edge.n_layers != 1 || d->edge_mlp.n_layers != 1 || d->node_mlp.n_layers != 1) return kFamilyGeneric;
    if (d->enc_node.n_layers < 1) return kFamilyGeneric;
    for (int i = 0; i < d->enc_node.n_layers; ++i)
        if (d->enc_node.layers[i].out_dim > 1024) return kFamilyGeneric;
    if (d->cls_edge.n_layers > 2) return kFamilyGeneric;
    if (d->cls_edge.n_layers == 2 && d->cls_edge.layers[0].out_dim > kMaxCls) return kFamilyGeneric;
    return kFamilyMfma32x6;
}

const gnncca_mlp& mlp_by_index(const gnncca_mpn_dims* d, int i) {
    switch (i) {
        case 0: return d->enc_node;
        case 1: return d->enc_edge;
        case 2: return d->edge_mlp;
        case 3: return d->node_mlp;
        default: return d->cls_edge;
    }
}

struct GenBlobPlan {
    GenBlobHeader h;
    size_t total_floats;
};

// layers of the fused step's weight block in stage order: fn(mlp index, layer, first row, rows, padded width)
template <typename F>
static void gen_step_block(const gnncca_mpn_dims* d, F fn) {
    const int hin = (d->reattach_nodes ? 2 : 1) * d->node_dim, ein = (d->reattach_edges ? 2 : 1) * d->edge_dim;
    const int mlps[4] = {2, 3, 4, 1};   // edge MLP, node MLP, classifier, then the edge ENCODER (step 1 runs it on the raw edge attributes)
    for (int m : mlps) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) {
            const int op = (mlp.layers[l].out_dim + 7) / 8 * 8;
            int k0 = 0, kn = mlp.layers[l].in_dim;
            if (l == 0 && m == 2) k0 = 2 * hin, kn = ein;            // cat[x[row] | x[col] | e] (models/mpn.py:68)
            if (l == 0 && m == 3) k0 = hin, kn = d->edge_dim;        // cat[x[row] | e'] (models/mpn.py:97)
            fn(m, l, k0, kn, op);
        }
    }
}

static GenBlobPlan plan_gen_blob(const gnncca_mpn_dims* d) {
    GenBlobPlan p;
    std::memset(&p, 0, sizeof(p));
    size_t off = align_up(sizeof(GenBlobHeader), 16) / 4;
    auto take = [&](size_t n) { size_t o = off; off = align_up(off + n, 4); return (int32_t)o; };
    p.h.magic = kBlobMagic;
    p.h.abi_version = GNNCCA_ABI_VERSION;
    p.h.family = kFamilyGeneric;
    for (int m = 0; m < 5; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) {
            // transposed and padded: [in][ceil8(out)] and [ceil8(out)] -- gen_dense_kernel reads 8 outputs per thread
            const size_t op = ((size_t)mlp.layers[l].out_dim + 7) / 8 * 8;
            p.h.w[m][l] = take((size_t)mlp.layers[l].in_dim * op);
            p.h.b[m][l] = take(op);
        }
    }
    // a wide first encoder layer (the 2048-d CNN embedding) runs on the MFMA family's split-K GEMM: keep its weight
    // row-major too
    if (d->enc_node.n_layers > 0 && d->enc_node.layers[0].in_dim >= 256 && d->enc_node.layers[0].in_dim % 64 == 0)
        p.h.enc0_rowmajor = take((size_t)d->enc_node.layers[0].in_dim * d->enc_node.layers[0].out_dim);
    // the fused step kernel (generic_fused.cuh) stages, per workgroup, the weights every EDGE uses: of the edge and node MLPs' first layers only
    // the rows of the e block (the x[row] / x[col] blocks became per-node projection tables), every further layer whole, the classifier,
    // each followed by its bias -- here once more, contiguous and in that order, so that the stage is ONE coalesced copy
    if (d->edge_mlp.n_layers > 0 && d->node_mlp.n_layers > 0) {
        size_t n = 0;
        gen_step_block(d, [&](int, int, int, int kn, int op) { n += (size_t)(kn + 1) * op; });
        p.h.step_w = take(n);
        p.h.step_w_floats = (int32_t)n;
    }
    p.total_floats = off;
    p.h.total_floats = (uint32_t)off;
    return p;
}

int gen_enc0_ksplit(const gnncca_mpn_dims* d, int64_t n_nodes) {
    if (classify(d) != kFamilyGeneric || plan_gen_blob(d).h.enc0_rowmajor == 0) return 0;
    const int k0 = d->enc_node.layers[0].in_dim;
    const size_t row_tiles = ((size_t)(n_nodes > 0 ? n_nodes : 0) + 31) / 32;
    int ks = 1;
    while (ks < 32 && row_tiles * (size_t)ks < 512 && k0 / (ks * 2) >= 64) ks *= 2;
    return ks;
}

bool gen_blob_header(const gnncca_mpn_dims* d, GenBlobHeader* out) {
    if (classify(d) != kFamilyGeneric || !out) return false;
    *out = plan_gen_blob(d).h;
    return true;
}

bool fast_consts_ok(const gnncca_mpn_dims* d) {
    return classify(d) == kFamilyMfma32x6 && d->edge_in == 4 && !d->reattach_nodes && !d->reattach_edges &&
           d->cls_edge.n_layers == 2 && d->cls_edge.layers[0].out_dim == 4;
}

bool enc_split_ok(const gnncca_mpn_dims* d) {
    return classify(d) == kFamilyMfma32x6 && d->enc_node.n_layers >= 2 && d->enc_node.layers[0].out_dim == 128 &&
           d->enc_node.layers[0].in_dim % 32 == 0;
}

// fp32 -> bf16, round to nearest even (finite inputs; weights are finite)
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// fp32 -> fp16, round to nearest even, subnormals kept, |f| >= 65520 -> infinity: bit for bit v_cvt_f16_f32 in the default float mode
// (what `(_Float16)v` compiles to in pack_device_kernel)
static inline uint16_t f16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((a > 0x7F800000u) ? 0x0200u : 0u));   // inf / NaN
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                          // >= 65520: rounds to infinity
    if (a >= 0x38800000u) {                                                                            // normal fp16 range: 2^-14 ...
        const uint32_t m = a - 0x38000000u;                                                            // rebias 127 -> 15
        return (uint16_t)(sign | (uint16_t)((m + 0xFFFu + ((m >> 13) & 1u)) >> 13));                  // (a carry into the exponent is the right answer)
    }
    if (a < 0x33000000u) return sign;                                                                  // < 2^-25: rounds to zero (2^-25 itself ties to even = 0)
    // subnormal fp16: value = mant * 2^-24 with mant = round(|f| * 2^24)
    const int e = (int)(a >> 23);                                                                      // 102 ... 112
    const uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;                                                // 24-bit significand, |f| = mant * 2^(e - 150)
    const int sh = 126 - e;                                                                            // mant >> sh = |f| * 2^24
    const uint32_t q = mant >> sh, rem = mant & ((1u << sh) - 1u), half = 1u << (sh - 1);
    return (uint16_t)(sign | (uint16_t)(q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
}
static inline float f16_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0x1Fu) u = sign | 0x7F800000u | (m << 13);
    else if (e != 0) u = sign | ((e + 112u) << 23) | (m << 13);
    else if (m == 0) u = sign;
    else {
        int s = 0;
        uint32_t mm = m;
        while (!(mm & 0x400u)) mm <<= 1, ++s;
        u = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3FFu) << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// position (in halfs) of piece 0 of weight element (output column o, input i) in the enc_w2h image; piece 1 sits 128 * 32 halfs further.
// FRAGMENT-major inside a 32-deep chunk: [piece][column tile c = o / 32][k-step s = (i % 32) / 16][lane = o % 32 + 32 * ((i % 16) / 8)][i % 8] --
// the 1 KB the 64 lanes of one B operand of v_mfma_f32_32x32x16_f16 hold is one contiguous run, so the same image serves the LDS ring of
// the 256-row kernel (ds_read_b128 at consecutive addresses: conflict-free without a swizzle) and the straight-to-register loads of the
// 32-row kernel (one fully coalesced 1 KB load per fragment)
static inline size_t w2h_index(int o, int i) {
    const int kk = i % 32, lane = (o % 32) + 32 * ((kk % 16) / 8);
    return (size_t)(i / 32) * (2 * 128 * 32) + (size_t)((((o / 32) * 2 + kk / 16) * 64 + lane) * 8 + (kk % 8));
}

// ---------------------------------------------------------------------------------------------
struct BlobPlan {
    BlobHeader h;
    size_t total_floats;
};

static BlobPlan plan_blob(const gnncca_mpn_dims* d) {
    BlobPlan p;
    std::memset(&p, 0, sizeof(p));
    size_t off = align_up(sizeof(BlobHeader), 16) / 4;
    auto take = [&](size_t n) { size_t o = off; off = align_up(off + n, 4); return (int32_t)o; };
    const int nf = d->reattach_nodes ? 2 : 1, ef = d->reattach_edges ? 2 : 1;
    p.h.magic = kBlobMagic;
    p.h.abi_version = GNNCCA_ABI_VERSION;
    p.h.family = kFamilyMfma32x6;
    p.h.enc_node_layers = d->enc_node.n_layers;
    for (int i = 0; i < d->enc_node.n_layers; ++i) {
        const gnncca_layer& l = d->enc_node.layers[i];
        p.h.enc_node_w[i] = take((size_t)l.in_dim * l.out_dim);
        p.h.enc_node_b[i] = take(l.out_dim);
    }
    const gnncca_layer& last = d->enc_node.layers[d->enc_node.n_layers - 1];
    p.h.enc_last_wT = take((size_t)last.in_dim * kH);
    p.h.enc_edge_w = take((size_t)kEF * d->edge_in);
    p.h.enc_edge_b = take(kEF);
    p.h.wee = take((size_t)kEF * ef * kEF);
    p.h.wne_b = take(3 * 64);
    p.h.wne_bf16 = take(9 * 64);
    p.h.proj_wT = take((size_t)nf * kH * kProjOut);
    p.h.proj_b = take(kProjOut);
    p.h.cls_layers = d->cls_edge.n_layers;
    p.h.cls_hidden = d->cls_edge.n_layers == 2 ? d->cls_edge.layers[0].out_dim : 0;
    if (d->cls_edge.n_layers == 2) {
        p.h.cls_w1 = take((size_t)p.h.cls_hidden * kEF);
        p.h.cls_b1 = take(p.h.cls_hidden);
        p.h.cls_w2 = take(p.h.cls_hidden);
        p.h.cls_b2 = take(1);
    } else {
        p.h.cls_w1 = take(kEF);
        p.h.cls_b1 = take(1);
        p.h.cls_w2 = p.h.cls_b2 = 0;
    }
    p.h.fast_consts = fast_consts_ok(d) ? take(kFastConsts) : 0;
    p.h.enc_w3 = enc_split_ok(d) ? take((size_t)3 * d->enc_node.layers[0].in_dim * d->enc_node.layers[0].out_dim / 2) : 0;
    if (p.h.enc_w3) {   // (enc_split_ok: out = 128) the same weight as two fp16 pieces for enc_f16.cuh, plus its overflow words
        p.h.enc_w2h = take((size_t)d->enc_node.layers[0].in_dim * d->enc_node.layers[0].out_dim);   // 2 pieces x 2 B = 4 B per element
        p.h.enc_w2h_bad = take(kW2hBadWords);
    }
    p.total_floats = off;
    p.h.total_floats = (uint32_t)off;
    return p;
}

bool blob_header(const gnncca_mpn_dims* d, BlobHeader* out) {
    if (classify(d) != kFamilyMfma32x6 || !out) return false;
    *out = plan_blob(d).h;
    return true;
}

// One Linear with its BatchNorm folded in: returns W' [out][in] and b' [out].
struct Folded {
    std::vector<float> w, b;
    int in, out;
};

static Folded fold_layer(const gnncca_layer& l, const float* const*& cur) {
    Folded f;
    f.in = l.in_dim;
    f.out = l.out_dim;
    const float* w = *cur++;
    const float* b = *cur++;
    f.w.resize((size_t)l.in_dim * l.out_dim);
    f.b.resize(l.out_dim);
    if (!l.has_bn) {
        std::memcpy(f.w.data(), w, f.w.size() * sizeof(float));
        std::memcpy(f.b.data(), b, f.b.size() * sizeof(float));
        return f;
    }
    const float* gamma = *cur++;
    const float* beta = *cur++;
    const float* mean = *cur++;
    const float* var = *cur++;
    for (int o = 0; o < l.out_dim; ++o) {
        // BatchNorm1d eval, eps = 1e-5 (torch default; models/mlp.py:15)
        const double s = (double)gamma[o] / std::sqrt((double)var[o] + 1e-5);
        for (int i = 0; i < l.in_dim; ++i) f.w[(size_t)o * l.in_dim + i] = (float)((double)w[(size_t)o * l.in_dim + i] * s);
        f.b[o] = (float)(((double)b[o] - (double)mean[o]) * s + (double)beta[o]);
    }
    return f;
}


// ---------------------------------------------------------------------------------------------
// The blob of gnncca_pack_weights restated as data: every region is a strided copy out of one parameter tensor,
// optionally through the BatchNorm fold.  Kept next to the host packer on purpose -- tests/test_boundary.py and
// tests/test_gpu_parity.py check that interpreting this program reproduces the host blob byte for byte.
bool pack_program(const gnncca_mpn_dims* d, PackProgram* out) {
    if (!out || classify(d) != kFamilyMfma32x6) return false;
    const BlobPlan p = plan_blob(d);
    std::memset(out, 0, sizeof(*out));
    out->header = p.h;
    int n = 0;
    bool overflow = false;
    struct Lin {
        int w, b, bn, in, out;
    };
    int cur = 0;
    auto next = [&](const gnncca_layer& l) {
        Lin r;
        r.w = cur++;
        r.b = cur++;
        r.bn = l.has_bn ? cur : -1;
        if (l.has_bn) cur += 4;
        r.in = l.in_dim;
        r.out = l.out_dim;
        return r;
    };
    auto weight = [&](const Lin& L, int dst, int src_off, int rows, int cols, int unit0, int drs, int dcs, int srs, int scs) {
        if (n >= kMaxPackSegs) { overflow = true; return; }
        PackSeg& g = out->segs[n++];
        g.kind = 0, g.dst = dst, g.param = L.w, g.bn = L.bn, g.src_off = src_off, g.rows = rows, g.cols = cols, g.unit0 = unit0;
        g.drs = drs, g.dcs = dcs, g.srs = srs, g.scs = scs;
    };
    auto bias = [&](const Lin& L, int dst, int rows) {
        if (n >= kMaxPackSegs) { overflow = true; return; }
        PackSeg& g = out->segs[n++];
        g.kind = 1, g.dst = dst, g.param = L.b, g.bn = L.bn, g.rows = rows, g.cols = 1, g.drs = 1, g.srs = 1;
    };
    const int nf = d->reattach_nodes ? 2 : 1, ef = d->reattach_edges ? 2 : 1;
    const int hin = nf * kH, ein = ef * kEF;
    // encoder.node_mlp
    for (int i = 0; i < d->enc_node.n_layers; ++i) {
        const Lin L = next(d->enc_node.layers[i]);
        weight(L, p.h.enc_node_w[i], 0, L.out, L.in, 0, L.in, 1, L.in, 1);
        bias(L, p.h.enc_node_b[i], L.out);
        if (i == 0 && p.h.enc_w3) {
            if (n >= kMaxPackSegs) return false;
            PackSeg& g = out->segs[n++];
            g.kind = 2, g.dst = p.h.enc_w3, g.param = L.w, g.bn = L.bn, g.rows = L.out, g.cols = L.in, g.drs = 0, g.dcs = 0;
            g.srs = L.in, g.scs = 1, g.plane = L.out * 32;  // [in/32][3 pieces][out][32]: see gnncca_pack_weights
        }
        if (i == 0 && p.h.enc_w2h) {
            if (n >= kMaxPackSegs) return false;
            PackSeg& g = out->segs[n++];
            g.kind = 4, g.dst = p.h.enc_w2h, g.param = L.w, g.bn = L.bn, g.rows = L.out, g.cols = L.in, g.drs = 0, g.dcs = 0;
            g.srs = L.in, g.scs = 1, g.plane = p.h.enc_w2h_bad;
        }
        if (i == d->enc_node.n_layers - 1) weight(L, p.h.enc_last_wT, 0, kH, L.in, 0, 1, kH, L.in, 1);
    }
    // encoder.edge_mlp
    const Lin Le0 = next(d->enc_edge.layers[0]);
    weight(Le0, p.h.enc_edge_w, 0, kEF, d->edge_in, 0, d->edge_in, 1, d->edge_in, 1);
    bias(Le0, p.h.enc_edge_b, kEF);
    // MPNet.edge_model.edge_mlp : columns [src | dst | edge]   (mpn.py:68)
    const Lin Le = next(d->edge_mlp.layers[0]);
    weight(Le, p.h.proj_wT + 8, 0, kEF, hin, 0, 1, kProjOut, Le.in, 1);        // P_src
    weight(Le, p.h.proj_wT + 0, hin, kEF, hin, 0, 1, kProjOut, Le.in, 1);      // P_dst
    weight(Le, p.h.wee, 2 * hin, kEF, ein, 0, ein, 1, Le.in, 1);
    bias(Le, p.h.proj_b + 8, kEF);
    // MPNet.node_model.node_mlp : columns [x[row] | edge]        (mpn.py:97)
    const Lin Ln = next(d->node_mlp.layers[0]);
    weight(Ln, p.h.proj_wT + 16, 0, kH, hin, 0, 1, kProjOut, Ln.in, 1);        // Q
    bias(Ln, p.h.proj_b + 16, kH);
    for (int s = 0; s < 3; ++s)
        for (int h = 0; h < 2; ++h) weight(Ln, p.h.wne_b + s * 64 + h * 32, hin + 2 * s + h, kH, 1, 0, 1, 0, Ln.in, 0);
    {   // the same six columns as bf16 pieces in the lane layout of the message MFMAs' B operands (kind 3)
        if (n >= kMaxPackSegs) return false;
        PackSeg& g = out->segs[n++];
        g.kind = 3, g.dst = p.h.wne_bf16, g.param = Ln.w, g.bn = Ln.bn, g.src_off = hin, g.rows = kH, g.cols = kEF, g.unit0 = 0;
        g.drs = 0, g.dcs = 0, g.srs = Ln.in, g.scs = 1;
    }
    // classifier.edge_mlp
    const Lin Lc1 = next(d->cls_edge.layers[0]);
    weight(Lc1, p.h.cls_w1, 0, Lc1.out, kEF, 0, kEF, 1, kEF, 1);
    bias(Lc1, p.h.cls_b1, Lc1.out);
    Lin Lc2 = Lc1;
    if (d->cls_edge.n_layers == 2) {
        Lc2 = next(d->cls_edge.layers[1]);
        weight(Lc2, p.h.cls_w2, 0, 1, Lc2.in, 0, Lc2.in, 1, Lc2.in, 1);
        bias(Lc2, p.h.cls_b2, 1);
    }
    if (p.h.fast_consts) {  // the transposed copies mpn_step_fast_kernel reads into SGPR pairs
        const int fc = p.h.fast_consts;
        weight(Le0, fc + kFcEncW, 0, kEF, 4, 0, 1, kEF, 4, 1);
        bias(Le0, fc + kFcEncB, kEF);
        weight(Le, fc + kFcWee, 2 * hin, kEF, kEF, 0, 1, kEF, Le.in, 1);
        weight(Lc1, fc + kFcCw1, 0, 4, kEF, 0, 1, 4, kEF, 1);
        bias(Lc1, fc + kFcCb1, 4);
        weight(Lc2, fc + kFcCw2, 0, 1, 4, 0, 4, 1, 4, 1);
        bias(Lc2, fc + kFcCb2, 1);
        bias(Le, fc + kFcProjB + 8, kEF);
        bias(Ln, fc + kFcProjB + 16, kH);
    }
    out->n_segs = n;
    return !overflow && cur == gnncca_param_count(d);
}

}  // namespace gnncca

using namespace gnncca;

extern "C" {

int gnncca_abi_version(void) { return GNNCCA_ABI_VERSION; }

const char* gnncca_status_string(int status) {
    switch (status) {
        case GNNCCA_OK: return "ok";
        case GNNCCA_ERR_INVALID_ARG: return "invalid argument";
        case GNNCCA_ERR_UNSUPPORTED: return "GRAPH_NET_PARAMS not supported by this build's HIP kernels";
        case GNNCCA_ERR_WORKSPACE: return "workspace too small";
        case GNNCCA_ERR_HIP: return "HIP runtime error";
        case GNNCCA_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

int gnncca_param_count(const gnncca_mpn_dims* d) {
    if (!dims_valid(d)) return -1;
    return mlp_param_count(d->enc_node) + mlp_param_count(d->enc_edge) + mlp_param_count(d->edge_mlp) +
           mlp_param_count(d->node_mlp) + mlp_param_count(d->cls_edge);
}

int gnncca_supported(const gnncca_mpn_dims* d) {
    if (!dims_valid(d)) return GNNCCA_ERR_INVALID_ARG;
    return classify(d) == kFamilyNone ? GNNCCA_ERR_UNSUPPORTED : GNNCCA_OK;
}

int gnncca_num_outputs(const gnncca_mpn_dims* d) {
    if (!dims_valid(d)) return -1;
    if (d->num_enc_steps == 0) return 1;                       // mpn.py:295-297
    const int first = d->num_enc_steps - d->num_class_steps + 1;  // mpn.py:277
    int n = 0;
    for (int s = 1; s <= d->num_enc_steps; ++s) n += (s >= first);
    return n;
}

size_t gnncca_packed_weights_bytes(const gnncca_mpn_dims* d) {
    const Family fam = classify(d);
    if (fam == kFamilyNone) return 0;
    if (fam == kFamilyGeneric) return plan_gen_blob(d).total_floats * sizeof(float);
    return plan_blob(d).total_floats * sizeof(float);
}

static int pack_generic(const gnncca_mpn_dims* d, const float* const* params, void* packed_host, size_t packed_bytes);

int gnncca_pack_weights(const gnncca_mpn_dims* d, const float* const* params, int n_params, void* packed_host,
                        size_t packed_bytes) {
    if (!dims_valid(d) || !params || !packed_host) return GNNCCA_ERR_INVALID_ARG;
    if (classify(d) == kFamilyNone) return GNNCCA_ERR_UNSUPPORTED;
    if (n_params != gnncca_param_count(d)) return GNNCCA_ERR_INVALID_ARG;
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) return GNNCCA_ERR_INVALID_ARG;
    if (classify(d) == kFamilyGeneric) return pack_generic(d, params, packed_host, packed_bytes);
    const BlobPlan p = plan_blob(d);
    if (packed_bytes < p.total_floats * sizeof(float)) return GNNCCA_ERR_INVALID_ARG;
    float* blob = static_cast<float*>(packed_host);
    std::memset(blob, 0, p.total_floats * sizeof(float));
    std::memcpy(blob, &p.h, sizeof(BlobHeader));
    const int nf = d->reattach_nodes ? 2 : 1, ef = d->reattach_edges ? 2 : 1;
    const float* const* cur = params;

    // encoder.node_mlp
    for (int i = 0; i < d->enc_node.n_layers; ++i) {
        Folded f = fold_layer(d->enc_node.layers[i], cur);
        std::memcpy(blob + p.h.enc_node_w[i], f.w.data(), f.w.size() * sizeof(float));
        std::memcpy(blob + p.h.enc_node_b[i], f.b.data(), f.b.size() * sizeof(float));
        if (i == 0 && p.h.enc_w3) {
            // w = w0 + w1 + w2 with bf16 pieces (24 mantissa bits in all): the split-bf16 GEMM multiplies the
            // pieces on the bf16 MFMA pipe and recovers fp32-level accuracy
            // Layout [in/32][3 pieces][out][32]: the 32-deep k-chunk a GEMM workgroup stages per iteration (all three
            // pieces of all output columns, 24 KB for out = 128) is one contiguous run -- full-line coalesced loads.
            uint16_t* w3 = reinterpret_cast<uint16_t*>(blob + p.h.enc_w3);
            const size_t plane = (size_t)f.out * 32;
            for (int o = 0; o < f.out; ++o)
                for (int i = 0; i < f.in; ++i) {
                    const float v = f.w[(size_t)o * f.in + i];
                    const uint16_t h0 = bf16_rne(v);
                    const float r1 = v - bf16_to_float(h0);
                    const uint16_t h1 = bf16_rne(r1);
                    const float r2 = r1 - bf16_to_float(h1);
                    const size_t k = (size_t)(i / 32) * 3 * plane + (size_t)o * 32 + (size_t)(i % 32);
                    w3[k] = h0;
                    w3[plane + k] = h1;
                    w3[2 * plane + k] = bf16_rne(r2);
                }
        }
        if (i == 0 && p.h.enc_w2h) {
            // w = w0 + w1 / 2048 with fp16 pieces (w1 = the residual scaled into w0's exponent range; 22 significant bits in all): the
            // fp16-split GEMM of enc_f16.cuh needs THREE piece products where the bf16 form needs six.  A weight beyond fp16's range sets
            // the flag word of the pack block that owns the element (device packer: element t belongs to block (t / 256) % 64).
            uint16_t* w2 = reinterpret_cast<uint16_t*>(blob + p.h.enc_w2h);
            uint32_t* bad = reinterpret_cast<uint32_t*>(blob + p.h.enc_w2h_bad);
            for (int o = 0; o < f.out; ++o)
                for (int i = 0; i < f.in; ++i) {
                    const float v = f.w[(size_t)o * f.in + i];
                    const uint16_t h0 = f16_rne(v);
                    const float r = (v - f16_to_float(h0)) * 2048.0f;
                    const size_t k = w2h_index(o, i);
                    w2[k] = h0;
                    w2[k + 128 * 32] = f16_rne(r);
                    if (!(std::fabs(v) < kF16Limit)) bad[(((size_t)o * f.in + i) / 256) % kW2hBadWords] = 1u;
                }
        }
        if (i == d->enc_node.n_layers - 1)
            for (int o = 0; o < kH; ++o)
                for (int k = 0; k < f.in; ++k) blob[p.h.enc_last_wT + (size_t)k * kH + o] = f.w[(size_t)o * f.in + k];
    }
    // encoder.edge_mlp
    {
        Folded f = fold_layer(d->enc_edge.layers[0], cur);
        std::memcpy(blob + p.h.enc_edge_w, f.w.data(), f.w.size() * sizeof(float));
        std::memcpy(blob + p.h.enc_edge_b, f.b.data(), f.b.size() * sizeof(float));
    }
    // MPNet.edge_model.edge_mlp : columns [src nf*H | dst nf*H | edge ef*EF]   (mpn.py:68)
    const int hin = nf * kH, ein = ef * kEF;
    float* projT = blob + p.h.proj_wT;
    float* projb = blob + p.h.proj_b;
    {
        Folded f = fold_layer(d->edge_mlp.layers[0], cur);
        const int in = f.in;  // 2*hin + ein
        for (int o = 0; o < kEF; ++o) {
            for (int c = 0; c < hin; ++c) {
                projT[(size_t)c * kProjOut + 8 + o] = f.w[(size_t)o * in + c];        // P_src
                projT[(size_t)c * kProjOut + 0 + o] = f.w[(size_t)o * in + hin + c];  // P_dst
            }
            for (int g = 0; g < ein; ++g) blob[p.h.wee + (size_t)o * ein + g] = f.w[(size_t)o * in + 2 * hin + g];
            projb[8 + o] = f.b[o];
        }
    }
    // MPNet.node_model.node_mlp : columns [x[row] nf*H | edge EF]                (mpn.py:97)
    {
        Folded f = fold_layer(d->node_mlp.layers[0], cur);
        const int in = f.in;  // hin + EF
        for (int o = 0; o < kH; ++o) {
            for (int c = 0; c < hin; ++c) projT[(size_t)c * kProjOut + 16 + o] = f.w[(size_t)o * in + c];  // Q
            projb[16 + o] = f.b[o];
        }
        // B operand of v_mfma_f32_32x32x2_f32, k-step s: lane l holds B[k = l>>5][j = l&31] = Wne[j][2s + k]
        for (int s = 0; s < 3; ++s)
            for (int l = 0; l < 64; ++l) blob[p.h.wne_b + s * 64 + l] = f.w[(size_t)(l & 31) * in + hin + 2 * s + (l >> 5)];
        // B operands of the split-bf16 message (msg_bf16.cuh: MsgB): dword [plane * 3 + j][lane] = the pieces of
        // (W_ne[ch][2j], W_ne[ch][2j + 1]), ch = lane & 31; plane 0 = first pieces in both halves, plane 1 = second pieces,
        // plane 2 = third pieces in lanes < 32 and FIRST pieces in lanes >= 32
        uint16_t* wb = reinterpret_cast<uint16_t*>(blob + p.h.wne_bf16);
        for (int ch = 0; ch < kH; ++ch)
            for (int k = 0; k < kEF; ++k) {
                const float v = f.w[(size_t)ch * in + hin + k];
                const uint16_t h0 = bf16_rne(v);
                const float r1 = v - bf16_to_float(h0);
                const uint16_t h1 = bf16_rne(r1);
                const float r2 = r1 - bf16_to_float(h1);
                const uint16_t h2 = bf16_rne(r2);
                auto at = [&](int plane, int lane) { return ((size_t)(plane * 3 + k / 2) * 64 + lane) * 2 + (k & 1); };
                wb[at(0, ch)] = wb[at(0, ch + 32)] = h0;
                wb[at(1, ch)] = wb[at(1, ch + 32)] = h1;
                wb[at(2, ch)] = h2;
                wb[at(2, ch + 32)] = h0;
            }
    }
    // classifier.edge_mlp
    {
        Folded f1 = fold_layer(d->cls_edge.layers[0], cur);
        std::memcpy(blob + p.h.cls_w1, f1.w.data(), f1.w.size() * sizeof(float));
        std::memcpy(blob + p.h.cls_b1, f1.b.data(), f1.b.size() * sizeof(float));
        if (d->cls_edge.n_layers == 2) {
            Folded f2 = fold_layer(d->cls_edge.layers[1], cur);
            std::memcpy(blob + p.h.cls_w2, f2.w.data(), f2.w.size() * sizeof(float));
            std::memcpy(blob + p.h.cls_b2, f2.b.data(), f2.b.size() * sizeof(float));
        }
    }
    if (p.h.fast_consts) {  // contiguous copy of the per-step scalars for the specialised kernel
        float* fc = blob + p.h.fast_consts;
        auto transpose = [&](int dst, int src, int rows, int cols) {  // [rows][cols] -> [cols][rows]
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols; ++c) fc[dst + c * rows + r] = blob[src + r * cols + c];
        };
        transpose(kFcEncW, p.h.enc_edge_w, 6, 4);
        std::memcpy(fc + kFcEncB, blob + p.h.enc_edge_b, 6 * sizeof(float));
        transpose(kFcWee, p.h.wee, 6, 6);
        transpose(kFcCw1, p.h.cls_w1, 4, 6);
        std::memcpy(fc + kFcCb1, blob + p.h.cls_b1, 4 * sizeof(float));
        std::memcpy(fc + kFcCw2, blob + p.h.cls_w2, 4 * sizeof(float));
        std::memcpy(fc + kFcCb2, blob + p.h.cls_b2, 1 * sizeof(float));
        std::memcpy(fc + kFcProjB, blob + p.h.proj_b, kProjOut * sizeof(float));
    }
    return GNNCCA_OK;
}

static int pack_generic(const gnncca_mpn_dims* d, const float* const* params, void* packed_host, size_t packed_bytes) {
    const GenBlobPlan p = plan_gen_blob(d);
    if (packed_bytes < p.total_floats * sizeof(float)) return GNNCCA_ERR_INVALID_ARG;
    float* blob = static_cast<float*>(packed_host);
    std::memset(blob, 0, p.total_floats * sizeof(float));
    std::memcpy(blob, &p.h, sizeof(GenBlobHeader));
    const float* const* cur = params;
    for (int m = 0; m < 5; ++m) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) {
            Folded f = fold_layer(mlp.layers[l], cur);
            const size_t op = ((size_t)f.out + 7) / 8 * 8;  // padding stays zero
            for (int o = 0; o < f.out; ++o)
                for (int k = 0; k < f.in; ++k) blob[p.h.w[m][l] + (size_t)k * op + o] = f.w[(size_t)o * f.in + k];
            std::memcpy(blob + p.h.b[m][l], f.b.data(), f.b.size() * sizeof(float));
            if (m == 0 && l == 0 && p.h.enc0_rowmajor)
                std::memcpy(blob + p.h.enc0_rowmajor, f.w.data(), f.w.size() * sizeof(float));
        }
    }
    if (p.h.step_w) {   // the fused step's stage image: copies of the rows above
        size_t o = (size_t)p.h.step_w;
        gen_step_block(d, [&](int m, int l, int k0, int kn, int op) {
            std::memcpy(blob + o, blob + p.h.w[m][l] + (size_t)k0 * op, (size_t)kn * op * sizeof(float));
            o += (size_t)kn * op;
            std::memcpy(blob + o, blob + p.h.b[m][l], (size_t)op * sizeof(float));
            o += (size_t)op;
        });
    }
    return GNNCCA_OK;
}

}  // extern "C"

namespace gnncca {

// (The same invariant as carve()'s: flags / seg_ptr / col32 / perm of the generic family's workspace are read by gnncca_frames_forward's post
// stage after the forward -- fused generic step, op-by-op form and L == 0 alike.)
GenWorkspace carve_generic(const gnncca_mpn_dims* d, int64_t n, int64_t e) {
    GenWorkspace w;
    std::memset(&w, 0, sizeof(w));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t N = (size_t)(n > 0 ? n : 0), E = (size_t)(e > 0 ? e : 0);
    int64_t nw = d->node_dim, ew = d->edge_dim;
    for (int l = 0; l < d->enc_node.n_layers; ++l) nw = std::max<int64_t>(nw, d->enc_node.layers[l].out_dim);
    const int edge_side[4] = {1, 2, 3, 4};
    for (int m : edge_side) {
        const gnncca_mlp& mlp = mlp_by_index(d, m);
        for (int l = 0; l < mlp.n_layers; ++l) ew = std::max<int64_t>(ew, mlp.layers[l].out_dim);
    }
    if (d->reattach_nodes) nw = std::max<int64_t>(nw, 2 * d->node_dim);  // room for cat(initial, latent)
    if (d->reattach_edges) ew = std::max<int64_t>(ew, 2 * d->edge_dim);
    ew = std::max<int64_t>(ew, d->node_dim);  // per-edge messages [E][H]
    w.node_w = nw;
    w.edge_w = ew;
    w.flags = take(256);
    w.blockflags = take((E / 256 + 2) * 4);
    w.seg_ptr = take((N + 1) * 4);
    w.col32 = take(E * 4);
    w.perm = take(E * 4);
    w.cursor = take((N + 1) * 4);
    w.row32o = take(E * 4);
    w.col32o = take(E * 4);
    for (int i = 0; i < 3; ++i) w.node[i] = take(N * (size_t)nw * 4);
    w.h0 = take(N * (size_t)d->node_dim * 4);
    for (int i = 0; i < 4; ++i) w.edge[i] = take(E * (size_t)ew * 4);
    w.e0 = take(E * (size_t)d->edge_dim * 4);
    const int ks = gen_enc0_ksplit(d, n);
    w.partial = take(ks > 0 ? (size_t)ks * N * (size_t)d->enc_node.layers[0].out_dim * 4 : 0);
    {   // [N][2 * o1e + o1n]: P_src | P_dst | Q of the fused step kernel, ping-pong
        const size_t o1e = d->edge_mlp.n_layers > 0 ? (size_t)d->edge_mlp.layers[0].out_dim : 0;
        const size_t o1n = d->node_mlp.n_layers > 0 ? (size_t)d->node_mlp.layers[0].out_dim : 0;
        for (int i = 0; i < 2; ++i) w.tab[i] = take(N * (2 * o1e + o1n) * 4);
    }
    w.total = off;
    return w;
}

// The diagnostic switches this build reads.  With the gate open, a GNNCCA_* variable that is NOT in this list is
// reported once on stderr: a stale A/B script then says so in its log instead of comparing a build with itself.
static const char* const kDiagSwitches[] = {
    "GNNCCA_DIAG", "GNNCCA_LIB", "GNNCCA_STAMPS", "GNNCCA_STEP_R2", "GNNCCA_STEP_NOMEM", "GNNCCA_STEP_NOEPI", "GNNCCA_STEP_NOHOOK", "GNNCCA_STEP_EARLYBAR",
    "GNNCCA_STEP_NORANGE", "GNNCCA_RANGE_MAX_E", "GNNCCA_PD_LDS_MIN", "GNNCCA_PD_LDS_MAX", "GNNCCA_TAIL_NPW", "GNNCCA_WPS", "GNNCCA_NO_NT", "GNNCCA_NO_PAD",
    "GNNCCA_GEMM_DIRECT", "GNNCCA_GEMM_SPLIT_MIN", "GNNCCA_GEMM_LDS_MIN", "GNNCCA_GEMM_NOPIPE", "GNNCCA_NO_FUSE", "GNNCCA_NO_MFMA_TAIL",
    "GNNCCA_TAIL_MFMA_MIN", "GNNCCA_NO_RIDE", "GNNCCA_GEMM_DIRECT_WG", "GNNCCA_GEMM_R32_NST", "GNNCCA_GEN_UNFUSED", "GNNCCA_NPW", "GNNCCA_NPW_MIN_N", "GNNCCA_NPW_MIN_N_FIRST", "GNNCCA_DEFER_CLS", "GNNCCA_GEMM_X_L2_ROWS", "GNNCCA_NPW_MAX_CHUNKS", "GNNCCA_GEMM_KROT", "GNNCCA_GEMM_BF16", "GNNCCA_GEMM_F16_ARM", "GNNCCA_GEMM_F16_DIAG", "GNNCCA_GEMM_F16_PRIO", "GNNCCA_GEMM_R32F_MIN", "GNNCCA_GEMM_R32F_MAX", "GNNCCA_GEMM_F16_XNT", "GNNCCA_STEP_NT", "GNNCCA_COLNORM_NOLDS", "GNNCCA_POOL_BLOCKING", "GNNCCA_POOL_CHUNK", "GNNCCA_POOL_SPIN_US", "GNNCCA_GEMM_SLICES_MIN", "GNNCCA_GEMM_SLICES_MAX", "GNNCCA_GEMM_SLICES_NKS", "GNNCCA_GEMM_SLICES_WGS"};

extern "C" char** environ;

const char* diag_env(const char* name) {
    static const bool on = [] {
        const char* v = std::getenv("GNNCCA_DIAG");
        const bool open = v != nullptr && v[0] == '1' && v[1] == '\0';
        if (open && environ) {
            for (char** e = environ; *e; ++e) {
                if (std::strncmp(*e, "GNNCCA_", 7) != 0) continue;
                const char* eq = std::strchr(*e, '=');
                const size_t len = eq ? (size_t)(eq - *e) : std::strlen(*e);
                bool known = false;
                for (const char* k : kDiagSwitches) known = known || (std::strlen(k) == len && std::strncmp(k, *e, len) == 0);
                if (!known) {
                    std::fprintf(stderr, "[gnncca] GNNCCA_DIAG=1: '%.*s' is not a switch of this build and is ignored; recognised:", (int)len, *e);
                    for (const char* k : kDiagSwitches) std::fprintf(stderr, " %s", k);
                    std::fprintf(stderr, "\n");
                }
            }
        }
        return open;
    }();
    return on ? std::getenv(name) : nullptr;
}

// An integer-valued diagnostic switch: `fallback` when unset; a value outside [lo, hi] is reported once and ignored.
int diag_env_int(const char* name, int fallback, int lo, int hi) {
    const char* v = diag_env(name);
    if (!v) return fallback;
    char* end = nullptr;
    const long x = std::strtol(v, &end, 10);
    if (end == v || *end != '\0' || x < lo || x > hi) {
        std::fprintf(stderr, "[gnncca] GNNCCA_DIAG=1: %s=%s is outside [%d, %d] and is ignored (default %d)\n", name, v, lo, hi, fallback);
        return fallback;
    }
    return (int)x;
}

int ell_stride(const gnncca_mpn_dims* d, int64_t n, int64_t e) {
    static const bool disabled = diag_env("GNNCCA_NO_PAD") != nullptr;  // diagnostics: A/B against the compact layout
    if (disabled) return 0;
    if (!fast_consts_ok(d) || d->agg == GNNCCA_AGG_MAX || d->num_enc_steps < 1 || n <= 0) return 0;
    if (e < (1 << 19)) return 0;                           // small graphs are cache-resident and latency-bound: nothing to gain
    const int64_t avg = (e + n - 1) / n;
    const int64_t S = (avg + 31) / 32 * 32;                // whole 128-B lines per feature plane and node
    if ((double)e < 0.90 * (double)n * (double)S) return 0;  // ragged: padding would cost more traffic than the shared lines
                                                           // (200 x dense100, 77 % fill: 5 % slower padded)
    if (n * S >= (1ll << 31) - 64) return 0;
    return (int)S;
}

int enc_lds_ksplit(int64_t n_nodes, int K) {
    const int64_t b = (n_nodes + 255) / 256;               // row blocks
    // time in units of one un-split workgroup's run (~237 us): rounds of 256 workgroups / k, + the fused epilogue (un-split) or
    // the tail kernel's pass over the k slabs (measured: 14 us at N = 16 384 / k = 4, 76 us at 66 048 / k = 4).
    // N = 65 536 -> 1 (256 blocks, one round); 66 048 -> 4 (258 blocks would take two rounds un-split: 2.07 vs 1.25 + 0.33);
    // 51 200 -> 1 (200 blocks: 1.07 vs 2 / 2 + 0.15); 16 384 -> 4.
    int best = 1;
    double best_cost = (double)((b + 255) / 256) + 0.07;
    for (int k = 2; k <= 8; k *= 2) {
        if ((K / k) % 32 != 0) break;
        const double cost = (double)((b * k + 255) / 256) / k + (0.05 + 0.07 * k) * (double)b / 256.0;
        if (cost < best_cost - 1e-9) best = k, best_cost = cost;
    }
    return best;
}

// INVARIANT (gnncca_frames_forward, csrc/mpn_forward.hip): the plan regions -- flags, seg_ptr, col32, perm -- stay LIVE AND FINAL from the plan's
// launch(es) until the forward returns: the post stage reads them from this workspace (found by carving it again) instead of building the
// plan a second time.  Every forward route finishes that plan (the riding plan blocks + plan_finish in a tail / fused-epilogue workgroup, the
// plan-only launch of big batches, L == 0), and no step kernel may reuse those regions.  tests/test_gpu_pipeline.py holds one pipeline call
// against the separate calls on every route.
Workspace carve(const gnncca_mpn_dims* d, int64_t n, int64_t e) {
    Workspace w;
    std::memset(&w, 0, sizeof(w));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t N = (size_t)(n > 0 ? n : 0), E = (size_t)(e > 0 ? e : 0);
    w.e_stride = (int64_t)align_up(E > 0 ? E : 1, 64);
    w.ell_S = ell_stride(d, n, e);
    if (w.ell_S > 0) w.e_stride = std::max<int64_t>(w.e_stride, (int64_t)align_up(N * (size_t)w.ell_S, 64));
    // split-K of the first (widest) encoder GEMM: enough workgroups to cover the chip when N is small
    const int k0 = d->enc_node.layers[0].in_dim;
    const size_t row_tiles = (N + 31) / 32;
    int ks = 1;
    while (ks < 32 && row_tiles * (size_t)ks < 512 && k0 / (ks * 2) >= 64) ks *= 2;
    // from 1024 nodes on the split-bf16 GEMM runs with 128-row workgroups: room for its split-K factor too
    if (N >= 1024)
        while (ks < 8 && ((N + 127) / 128) * (size_t)ks < 512 && k0 / (ks * 2) >= 64) ks *= 2;
    if (N >= 4096) ks = std::max(ks, enc_lds_ksplit((int64_t)N, k0));  // room for the slabs of the 256-row GEMM's choice
    w.ksplit = ks;
    size_t fmax = 0;
    for (int i = 0; i < d->enc_node.n_layers; ++i)
        fmax = fmax > (size_t)d->enc_node.layers[i].out_dim ? fmax : (size_t)d->enc_node.layers[i].out_dim;
    w.flags = take(256);
    w.blockflags = take((E / 256 + 2) * 4);
    w.seg_ptr = take((N + 1) * 4);
    w.col32 = take(E * 4);
    w.perm = take(E * 4);
    w.cursor = take((N + 1) * 4);
    w.h0 = take(N * kH * 4);
    w.act = take(d->enc_node.n_layers >= 3 ? 2 * N * fmax * 4 : 0);
    w.partial = take((size_t)ks * N * fmax * 4);
    for (int i = 0; i < 2; ++i) w.pd[i] = take(N * kPdStride * 4);
    for (int i = 0; i < 2; ++i) w.psq[i] = take(N * kPsQStride * 4);
    w.e = take((size_t)kEF * w.e_stride * 4);
    w.e0 = take(d->reattach_edges ? (size_t)kEF * w.e_stride * 4 : 0);
    w.rng = take(N * 16);   // per-node column ranges (start1, len1, start2 - len1, breaks): written by step 1, read by the later steps
    w.total = off;
    return w;
}

}  // namespace gnncca

extern "C" size_t gnncca_workspace_bytes(const gnncca_mpn_dims* d, int64_t n_nodes, int64_t n_edges) {
    const Family fam = classify(d);
    if (fam == kFamilyNone || n_nodes < 0 || n_edges < 0) return 0;
    if (fam == kFamilyGeneric) return carve_generic(d, n_nodes, n_edges).total;
    return carve(d, n_nodes, n_edges).total;
}
