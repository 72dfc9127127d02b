#pragma once
// SHELVED EXPERIMENT (round 3; not compiled): mpn_step_pipe_kernel inside persistent waves.  It was correct (115 GPU parity / fuzz
// tests) and, before the one-node-per-wave kernel had its VALU count cut, 6 % ahead of it on 64 x dense256 and 512 x dense128
// (profiles/r03_logs/r3_abc1.log); after the cut the two tie there and the persistent form loses 11 % at 200 x dense256
// (profiles/r03_logs/r3_ab_persist2.log: 176 vs 157 us per launch) -- with the kernel bound by VALU issue there is
// no idle time left for cross-node prefetch to fill, and fixed node-to-wave assignment costs balance.  Requesting a node's first
// four chunks at once spilled 21 registers at four waves per SIMD (46 -> 67 us; r3_abc3_request4.log).
// To revive: paste back into csrc/step_pipe.cuh before launch_pipe_t and dispatch on N >= 16 x kPersistWgPerCu x CUs.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// The same kernel with PERSISTENT waves, for batches with more segments than resident waves (N > 16 x CUs; the gather
// table is then far beyond LDS anyway).  Round-3 stamps of the one-node-per-wave form on 64 x dense256
// (profiles/r03_logs/r3_stamps_64x256.log): a wave lived 14.4 us, 2.5 of them before its first load left (flag word, CSR
// offsets, descriptors, message weights, projection table staging), 1.3 at the workgroup barrier, 1.2-1.4 in the
// projection epilogue while nothing of the NEXT node was in flight -- and a dense-128 node has one round to amortise all
// of that over.  Here wave w walks nodes w, w + W, w + 2W, ... (W = every resident wave of the launch): the projection
// table is staged and the message weights are split ONCE per wave, there is no barrier after the first one, the next
// node's CSR offsets arrive by s_load one node ahead, and its first-round loads (target ids, edge state, the (P_src | Q)
// row) are issued BEFORE the current node's epilogue, so they fly while the segment sum is projected and stored.
// ------------------------------------------------------------------------------------------------------------
constexpr int kPersistWgPerCu = 4;

template <bool FIRST, bool CLS, bool MSG, bool EBF16, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void mpn_step_persist_kernel(const StepParams p) {
    constexpr bool PD_LDS = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_pd = nullptr;                                  // (named by the shared pieces; PD_LDS is off here)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ch = lane & 31;
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);
    // CSR offsets through the constant address space: a wave-uniform index becomes s_load, which can stay pending across a whole
    // node (a vector load would be waited for with vmcnt(0) the moment it is turned into SGPRs); written by the plan launch, not here
    typedef const int __attribute__((address_space(4))) cint;
    cint* seg_c = (cint*)(unsigned long long)p.seg_ptr;
    constexpr int AUX_ST = NT >= 1 ? 2 : 0;
    constexpr int AUX_LD = NT >= 2 ? 2 : 0;

    const unsigned gflags = p.flags[0];
    MsgB mb;
    float projb_l = 0.f;
    if (MSG) {   // once per workgroup / wave
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        for (int i = tid; i < kH * kProjOut / 4; i += 256) l4[i] = g4[i];
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
        msg_b_weights(blob + p.off_wnebf, lane, mb);
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    if (MSG) __syncthreads();
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    const bool padded = p.ell_S > 0 && !(gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR));
    const int W = (int)gridDim.x * 4;
    int node = (int)blockIdx.x * 4 + wave;
    if (node >= p.N) return;

    const unsigned plane_b = (unsigned)p.e_stride * 4u;
    const unsigned long long live = (p.diag & 1) ? 0ull : 1ull;
    const rsrc_t r_e = make_rsrc(p.e, live * (EBF16 ? 3 : 6) * plane_b);
    const rsrc_t r_col = make_rsrc(p.col32, live * (unsigned long long)p.E * 4);
    const rsrc_t r_perm = make_rsrc(p.perm, unsorted ? (unsigned long long)p.E * 4 : 0ull);
    const rsrc_t r_attr = make_rsrc(p.edge_attr, live * (unsigned long long)p.E * 16);
    const rsrc_t r_pd = make_rsrc(p.pd_in, (unsigned long long)p.N * (kPdStride * 4));
    const rsrc_t r_log = make_rsrc(p.logits, CLS ? live * (unsigned long long)p.E * 4 : 0ull);

    // per-node state the shared pieces read (by reference)
    int seg_s = seg_c[node], seg_t = seg_c[node + 1];
    int eoff = padded ? node * p.ell_S - seg_s : 0;
    float psrc[kEF];
    float cinit = 0.f;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

#define GNNCCA_MID_HOOK() do { } while (0)
#include "step_pieces.inc"
#undef GNNCCA_MID_HOOK

    auto load_row = [&](int n) {   // the node's (P_src | Q) row
        const float* __restrict__ psq = p.psq_in + (size_t)n * kPsQStride;
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
        if (MSG) cinit = psq[8 + ch];
    };
    // (Requesting a node's first FOUR chunks together -- ids and state depend on the CSR offsets only -- was measured: 21 spilled
    // registers at four waves per SIMD, 64 x dense256 46 -> 67 us per launch; profiles/r03_logs/r3_abc3_request4.log.)
    Chunk c0, c1;
    auto request_node = [&]() {
        load_index(seg_s, c0);
        load_index(seg_s + 64, c1);
        load_state(seg_s, c0);
        load_state(seg_s + 64, c1);
    };
    load_row(node);
    request_node();

    for (;;) {
        const int next = node + W;
        const bool has_next = next < p.N;
        int ns = 0, nt = 0;
        if (has_next) ns = seg_c[next], nt = seg_c[next + 1];   // s_load: lands while this node is computed
        if (MSG) msg_b_bias(cinit, lane, mb);
        auto round_body = [&](int rb, Chunk& a, Chunk& b) {
            load_target(a);
            load_target(b);
            if (rb + 64 < seg_t)
                compute2(rb, a, rb + 64, b);
            else
                compute1(rb, a);
        };
        int base = seg_s;
        if (base < seg_t) {
            // the SECOND round's target ids are requested before the first round is computed (see mpn_step_fast_kernel)
            const int base2 = base + 128;
            Chunk n0, n1;
            load_index(base2, n0);
            load_index(base2 + 64, n1);
            round_body(base, c0, c1);
            if (base2 < seg_t) {
                load_state(base2, n0);
                load_state(base2 + 64, n1);
                round_body(base2, n0, n1);
            }
            base += 256;
        }
        for (; base < seg_t; base += 128) {
            load_index(base, c0);
            load_index(base + 64, c1);
            load_state(base, c0);
            load_state(base + 64, c1);
            round_body(base, c0, c1);
        }
        const int deg = seg_t - seg_s;
        const int done = node;
        // ---- the next node's first round goes out before this node's epilogue -------------------------------------------------
        node = next;
        seg_s = ns, seg_t = nt;   // (0, 0 past the last node: every lane out of range, no traffic)
        eoff = (padded && has_next) ? node * p.ell_S - seg_s : 0;
        if (has_next) load_row(node);   // first: the bias operand of the next node's first MFMA waits for it alone
        request_node();
        if (MSG) {
            float v = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += acc[i];
            v += __shfl_xor(v, 32);
            if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
            if (deg == 0) v = 0.f;
            // projection epilogue (project_node with the bias read through the constant address space)
            const int o = min(lane, kProjOut - 1);
            float pr = projb_l;
            const float* w = s_proj + o;
#pragma unroll
            for (int c = 0; c < kH; ++c)
                pr = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)), pr);
            if (lane < kPdStride)
                p.pd_out[(size_t)done * kPdStride + lane] = pr;
            else if (lane < kProjOut)
                p.psq_out[(size_t)done * kPsQStride + lane - kPdStride] = pr;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
        if (!has_next) break;
    }
}

template <bool FIRST, bool CLS, bool MSG, bool EB, int NT>
static hipError_t launch_persist_t(const StepParams& sp, hipStream_t st, int n_cu) {
    const size_t lds = MSG ? (size_t)kH * kProjOut * sizeof(float) : 0;
    // persistent grid = exactly the workgroups that are resident at once (asked once per instantiation)
    static thread_local int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mpn_step_persist_kernel<FIRST, CLS, MSG, EB, NT>), 256,
                                                         lds) != hipSuccess || n < 1)
            n = 2;
        wg_per_cu = std::min(n, kPersistWgPerCu);
    }
    const unsigned blocks = (unsigned)std::min<long long>(((long long)sp.N + 3) / 4, (long long)n_cu * wg_per_cu);
    GNNCCA_LAUNCH((mpn_step_persist_kernel<FIRST, CLS, MSG, EB, NT>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool FIRST, bool CLS, bool MSG>
static hipError_t launch_persist(const StepParams& sp, hipStream_t st, int n_cu) {
    const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
    if (nt == 2) return sp.e_bf16 ? launch_persist_t<FIRST, CLS, MSG, true, 2>(sp, st, n_cu) : launch_persist_t<FIRST, CLS, MSG, false, 2>(sp, st, n_cu);
    if (nt == 1) return sp.e_bf16 ? launch_persist_t<FIRST, CLS, MSG, true, 1>(sp, st, n_cu) : launch_persist_t<FIRST, CLS, MSG, false, 1>(sp, st, n_cu);
    return sp.e_bf16 ? launch_persist_t<FIRST, CLS, MSG, true, 0>(sp, st, n_cu) : launch_persist_t<FIRST, CLS, MSG, false, 0>(sp, st, n_cu);
}

static hipError_t launch_persist_dispatch(const StepParams& sp, bool msg, hipStream_t st, int n_cu) {
    const int key = (sp.first ? 4 : 0) | (sp.cls_layers ? 2 : 0) | (msg ? 1 : 0);
    switch (key) {
        case 0: return launch_persist<false, false, false>(sp, st, n_cu);
        case 1: return launch_persist<false, false, true>(sp, st, n_cu);
        case 2: return launch_persist<false, true, false>(sp, st, n_cu);
        case 3: return launch_persist<false, true, true>(sp, st, n_cu);
        case 4: return launch_persist<true, false, false>(sp, st, n_cu);
        case 5: return launch_persist<true, false, true>(sp, st, n_cu);
        case 6: return launch_persist<true, true, false>(sp, st, n_cu);
        case 7: return launch_persist<true, true, true>(sp, st, n_cu);
    }
    return hipErrorInvalidValue;
}

}  // namespace gnncca
