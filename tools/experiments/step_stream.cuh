#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Step kernel for BATCHES (N > 1024 nodes: the gather table no longer fits LDS and there are many more segments than
// resident waves): the arithmetic of mpn_step_pipe_kernel inside PERSISTENT waves that stream their segments through
// wave-private LDS slots.
//
// Why (round-3 stamps of the one-node-per-wave form, 64 x dense256, profiles/r03_logs/r3_stamps_64x256.log): a wave lived
// 14.4 us of which 6.5 were arithmetic; the rest were dependent round trips a fresh wave cannot avoid -- CSR offsets and
// the node's (P_src | Q) row (2.5 us), the workgroup barrier behind the table staging (1.3), target ids (1.2), the P_dst
// gather (1.0), the second round's state (0.85) -- and the four waves of a SIMD, started together, sat in the same phase
// at the same time.  Here a wave keeps going: wave w owns nodes w, w + W, w + 2W, ... (W = all resident waves) and cuts
// them into ITEMS of 128 edges (one round).  While item k is computed, item k + 1's edge state and target ids are already
// on their way into the wave's other LDS slot (LDS-DMA: buffer_load_dwordx4 ... lds, no registers held by bytes in
// flight) and its P_dst rows on their way into registers; the projection table, the message weights and the buffer
// descriptors are set up once per wave instead of once per node; there is no barrier after the first one.
//
// LDS per wave: two slots of [6 planes x 128 edges | 128 target ids | the node's (P_src | Q) row]; a 16-byte DMA lane
// covers four consecutive edges of a plane, the two half-waves take two planes per instruction.  Lanes beyond the item
// carry the out-of-range offset (no traffic).  A slot may hold stale or neighbouring values in such positions; the
// edge update adds -inf to their pre-activation, so e' = max(x, 0) is exactly 0 whatever was there (NaN included).
// Unsorted graphs (never produced by the reference; the flag is only known on the device) take a plain loop over the
// same pieces inside this launch.
// ------------------------------------------------------------------------------------------------------------
constexpr int kItemEdges = 128;                    // edges per item: two 64-edge chunks = one round
constexpr int kSlotCol = 6 * kItemEdges * 4;       // 3072: the feature planes (step 1: the item's edge_attr rows) come first
constexpr int kSlotPsq = kSlotCol + 1024;          // target ids + 512 B the idle half-wave's lanes may write
constexpr int kSlotBytes = kSlotPsq + 256;         // + the node's 40-float (P_src | pad | Q) row
constexpr int kStreamWgPerCu = 4;

template <bool FIRST, bool CLS, bool MSG, bool EBF16, int NT>
__global__ __launch_bounds__(256) GNNCCA_FAST_ATTR void mpn_step_stream_kernel(const StepParams p) {
    constexpr bool PD_LDS = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                               // [32][48]   (MSG)
    char* s_slots = reinterpret_cast<char*>(smem + (MSG ? kH * kProjOut : 0));   // [4 waves][2][kSlotBytes]
    float* s_pd = nullptr;                                              // (named by the shared pieces; PD_LDS is off here)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ch = lane & 31;
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);
    constexpr int AUX_ST = NT >= 1 ? 2 : 0;
    constexpr int AUX_LD = NT >= 2 ? 2 : 0;

    const unsigned gflags = p.flags[0];
    // CSR offsets through the constant address space: wave-uniform indices become s_load (a vector load would have to be waited
    // for with vmcnt(0) the moment it is turned into SGPRs, draining every DMA in flight); written by the plan launch, not here
    typedef const int __attribute__((address_space(4))) cint;
    cint* seg_c = (cint*)(unsigned long long)p.seg_ptr;
    char* my = s_slots + wave * (2 * kSlotBytes);
    if (MSG) {   // the projection matrix, once per workgroup
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        for (int i = tid; i < kH * kProjOut / 4; i += 256) l4[i] = g4[i];
    }
    {            // no NaN patterns in positions a DMA never writes
        f32x4* z4 = reinterpret_cast<f32x4*>(my);
        for (int i = lane; i < 2 * kSlotBytes / 16; i += 64) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    MsgB mb;
    float projb_l = 0.f;
    if (MSG) {
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
        msg_b_weights(blob + p.off_wneb, lane, mb);
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    __syncthreads();
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    const bool padded = p.ell_S > 0 && !(gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR));
    const int W = (int)gridDim.x * 4;
    const int node0 = (int)blockIdx.x * 4 + wave;
    if (node0 >= p.N) return;

    const unsigned plane_b = (unsigned)p.e_stride * 4u;
    const unsigned long long live = (p.diag & 1) ? 0ull : 1ull;
    const rsrc_t r_e = make_rsrc(p.e, live * (EBF16 ? 3 : 6) * plane_b);
    const rsrc_t r_col = make_rsrc(p.col32, live * (unsigned long long)p.E * 4);
    const rsrc_t r_perm = make_rsrc(p.perm, unsorted ? (unsigned long long)p.E * 4 : 0ull);
    const rsrc_t r_attr = make_rsrc(p.edge_attr, live * (unsigned long long)p.E * 16);
    const rsrc_t r_pd = make_rsrc(p.pd_in, (unsigned long long)p.N * (kPdStride * 4));
    const rsrc_t r_psq = make_rsrc(p.psq_in, (unsigned long long)p.N * (kPsQStride * 4));
    const rsrc_t r_log = make_rsrc(p.logits, CLS ? live * (unsigned long long)p.E * 4 : 0ull);
    const rsrc_t r_est = make_rsrc(p.e, p.store_e ? live * (EBF16 ? 3 : 6) * plane_b : 0ull);

    // per-node state the shared pieces read (by reference)
    int seg_t = 0, eoff = 0;
    float psrc[kEF] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // P_dst rows of the NEXT item: requested in the middle of the current item's arithmetic (GNNCCA_MID_HOOK), from target ids that
    // were read out of the other slot at the top of the iteration -- no LDS read, hence no compiler-inserted vmcnt(0), in between
    struct Pd2 {
        float a[kEF], b[kEF];
    };
    Pd2 Pn;
    int jn_a = 0, jn_b = 0;
    bool has1 = false;
    auto gather_rows = [&](int ja, int jb, Pd2& o) {
        const unsigned va = (unsigned)ja * (kPdStride * 4u), vb_ = (unsigned)jb * (kPdStride * 4u);
        const u32x4 a4 = __builtin_amdgcn_raw_buffer_load_b128(r_pd, va, 0, 0);
        const u32x2 a2 = __builtin_amdgcn_raw_buffer_load_b64(r_pd, va + 16u, 0, 0);
        const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(r_pd, vb_, 0, 0);
        const u32x2 b2 = __builtin_amdgcn_raw_buffer_load_b64(r_pd, vb_ + 16u, 0, 0);
        o.a[0] = __uint_as_float(a4[0]), o.a[1] = __uint_as_float(a4[1]), o.a[2] = __uint_as_float(a4[2]), o.a[3] = __uint_as_float(a4[3]);
        o.a[4] = __uint_as_float(a2[0]), o.a[5] = __uint_as_float(a2[1]);
        o.b[0] = __uint_as_float(b4[0]), o.b[1] = __uint_as_float(b4[1]), o.b[2] = __uint_as_float(b4[2]), o.b[3] = __uint_as_float(b4[3]);
        o.b[4] = __uint_as_float(b2[0]), o.b[5] = __uint_as_float(b2[1]);
    };
#define GNNCCA_MID_HOOK()                          \
    do {                                           \
        if (has1) gather_rows(jn_a, jn_b, Pn);     \
    } while (0)
#include "step_pieces.inc"
#undef GNNCCA_MID_HOOK

    // end of a node: segment sum -> (mean) -> the projections of the next step
    auto node_epilogue = [&](int node, int deg) {
        if (!MSG) return;
        float v = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += acc[i];
        v += __shfl_xor(v, 32);
        if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
        if (deg == 0) v = 0.f;
        const int o = min(lane, kProjOut - 1);
        float pr = projb_l;
        const float* w = s_proj + o;
#pragma unroll
        for (int c = 0; c < kH; ++c)
            pr = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)), pr);
        if (lane < kPdStride)
            p.pd_out[(size_t)node * kPdStride + lane] = pr;
        else if (lane < kProjOut)
            p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    };

    if (unsorted) {   // correctness path: one chunk at a time through the permutation, no prefetch
        for (int node = node0; node < p.N; node += W) {
            const int s = seg_c[node];
            seg_t = seg_c[node + 1];
            const float* __restrict__ psq = p.psq_in + (size_t)node * kPsQStride;
#pragma unroll
            for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
            if (MSG) msg_b_bias(psq[8 + ch], lane, mb);
            for (int base = s; base < seg_t; base += 64) {
                Chunk c;
                load_index(base, c);
                load_state(c);
                load_target(c);
                compute1(base, c);
            }
            node_epilogue(node, seg_t - s);
        }
        return;
    }

    // ---- item iterator (wave-uniform) ------------------------------------------------------------------------------------
    int it_node = node0, it_s = seg_c[node0], it_t = seg_c[node0 + 1], it_base = it_s;   // the item fetched next
    int nn_s = 0, nn_t = 0;                                                                       // the segment one node further
    if (node0 + W < p.N) nn_s = seg_c[node0 + W], nn_t = seg_c[node0 + W + 1];
    auto advance = [&]() -> bool {
        if (it_base + kItemEdges < it_t) {
            it_base += kItemEdges;
            return true;
        }
        it_node += W;
        if (it_node >= p.N) return false;
        it_s = nn_s, it_t = nn_t, it_base = it_s;
        if (it_node + W < p.N) nn_s = seg_c[it_node + W], nn_t = seg_c[it_node + W + 1];
        return true;
    };
    // LDS-DMA of item (node, base in [s, t)) into slot q
    auto issue = [&](int q, int node, int base, int s, int t) {
        char* dst = my + q * kSlotBytes;
        const int cnt = t - base;
        const int l32 = lane & 31;
        if (FIRST) {   // the item's edge_attr rows, lane = edge
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = c * 64 + lane;
                const unsigned vo = e < cnt ? (unsigned)(base + e) * 16u : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_attr, (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, vo, 0, 0, AUX_ST);
            }
        } else {
            const int eo = padded ? node * p.ell_S - s : 0;
            const bool on = 4 * l32 < cnt;
            constexpr int NPL = EBF16 ? 3 : 6;
#pragma unroll
            for (int pp = 0; pp < (NPL + 1) / 2; ++pp) {   // two planes per instruction: lanes < 32 plane 2pp, lanes >= 32 plane 2pp + 1
                const int f = 2 * pp + half;
                const unsigned vo = (on && f < NPL) ? (unsigned)f * plane_b + (unsigned)(base + eo + 4 * l32) * 4u : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_e, (__attribute__((address_space(3))) void*)(dst + pp * 1024), 16, vo, 0, 0, AUX_LD);
            }
        }
        {
            const unsigned vo = (!half && 4 * l32 < cnt) ? (unsigned)(base + 4 * l32) * 4u : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_col, (__attribute__((address_space(3))) void*)(dst + kSlotCol), 16, vo, 0, 0, 0);
        }
        if (base == s) {   // first item of its node: the (P_src | Q) row rides along
            const unsigned vo = lane < kPsQStride ? (unsigned)(node * kPsQStride + lane) * 4u : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_psq, (__attribute__((address_space(3))) void*)(dst + kSlotPsq), 4, vo, 0, 0, 0);
        }
    };
    // target ids of an item whose DMA has landed in slot q; lanes beyond the item name the out-of-range row (the gather returns 0)
    auto read_ids = [&](int q, int base, int t, int& ja, int& jb) {
        const int* colp = reinterpret_cast<const int*>(my + q * kSlotBytes + kSlotCol);
        constexpr int kDeadRow = (int)(kOobOffset / (kPdStride * 4u));
        ja = (base + lane < t) ? colp[lane] : kDeadRow;
        jb = (base + 64 + lane < t) ? colp[64 + lane] : kDeadRow;
    };

    // ---- pipeline prologue ---------------------------------------------------------------------------------------------------
    int cur_node = it_node, cur_s = it_s, cur_t = it_t, cur_base = it_base;
    issue(0, cur_node, cur_base, cur_s, cur_t);
    has1 = advance();
    int nx_node = it_node, nx_s = it_s, nx_t = it_t, nx_base = it_base;
    if (has1) issue(1, nx_node, nx_base, nx_s, nx_t);
    Pd2 P;
    {
        int ja, jb;
        read_ids(0, cur_base, cur_t, ja, jb);
        gather_rows(ja, jb, P);
        // drained here, once: the loop must not inherit a pending load of P from this path (hipcc would then wait for it with a
        // count that ignores the DMAs issued in between, inside every iteration)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(P.a[0]), "+v"(P.a[1]), "+v"(P.a[2]), "+v"(P.a[3]), "+v"(P.a[4]), "+v"(P.a[5]), "+v"(P.b[0]), "+v"(P.b[1]),
                       "+v"(P.b[2]), "+v"(P.b[3]), "+v"(P.b[4]), "+v"(P.b[5])
                     :
                     : "memory");
    }

    for (int k = 0;; ++k) {
        const int q = k & 1;
        const char* src = my + q * kSlotBytes;
        // item `cur` sits in slot q, its P_dst rows in P; item `nx` is landing in slot q ^ 1
        Chunk a, b;
        {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Chunk& x = c ? b : a;
                const int e = c * 64 + lane;
                if (FIRST) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + e * 16);
                    x.raw[0] = v[0], x.raw[1] = v[1], x.raw[2] = v[2], x.raw[3] = v[3], x.raw[4] = 0.f, x.raw[5] = 0.f;
                } else if (EBF16) {
#pragma unroll
                    for (int f = 0; f < kEF / 2; ++f) {
                        const unsigned w = *reinterpret_cast<const unsigned*>(src + f * 512 + e * 4);
                        x.raw[2 * f] = __uint_as_float(w << 16);
                        x.raw[2 * f + 1] = __uint_as_float(w & 0xFFFF0000u);
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < kEF; ++f) x.raw[f] = *reinterpret_cast<const float*>(src + f * 512 + e * 4);
                }
            }
        }
        if (cur_base == cur_s) {   // a new node: its (P_src | Q) row, fresh accumulators (node_epilogue cleared them)
            const float* psq = reinterpret_cast<const float*>(src + kSlotPsq);
#pragma unroll
            for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
            if (MSG) msg_b_bias(psq[8 + ch], lane, mb);
        }
        if (has1) read_ids(q ^ 1, nx_base, nx_t, jn_a, jn_b);
        // the reads of slot q are complete before the next DMA may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool has2 = has1 && advance();
        if (has2) issue(q, it_node, it_base, it_s, it_t);

        // ---- arithmetic of item `cur` -----------------------------------------------------------------------------------
        seg_t = cur_t;
        const int eo = padded ? cur_node * p.ell_S - cur_s : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Chunk& x = c ? b : a;
            const int kk = cur_base + c * 64 + lane;
            const bool valid = kk < cur_t;
            x.vo_e = valid ? (unsigned)(kk + eo) * 4u : kOobOffset;
            x.vo_out = valid ? (unsigned)kk * 4u : kOobOffset;
            x.vo_attr = kOobOffset;
            const float vb = valid ? 0.f : -__builtin_inff();   // a dead lane's pre-activation is -inf (+ anything, NaN included): e' = 0
#pragma unroll
            for (int f = 0; f < kEF; ++f) x.pd[f] = (c ? P.b[f] : P.a[f]) + vb;
        }
        if (cur_base + 64 < cur_t)
            compute2(cur_base, a, cur_base + 64, b);
        else if (cur_base < cur_t)
            compute1(cur_base, a);
        else if (has1)
            gather_rows(jn_a, jn_b, Pn);   // a node without edges
        if (cur_base + kItemEdges >= cur_t) node_epilogue(cur_node, cur_t - cur_s);
        if (!has1) break;
        cur_node = nx_node, cur_s = nx_s, cur_t = nx_t, cur_base = nx_base;
        nx_node = it_node, nx_s = it_s, nx_t = it_t, nx_base = it_base;
        P = Pn;
        has1 = has2;
    }
}

template <bool FIRST, bool CLS, bool MSG, bool EB, int NT>
static hipError_t launch_stream_t(const StepParams& sp, hipStream_t st, int n_cu) {
    const size_t lds = (MSG ? (size_t)kH * kProjOut * sizeof(float) : 0) + (size_t)4 * 2 * kSlotBytes;
    // persistent grid = exactly the workgroups that are resident at once (registers and LDS decide: asked once per instantiation)
    static thread_local int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mpn_step_stream_kernel<FIRST, CLS, MSG, EB, NT>), 256, lds) != hipSuccess || n < 1) n = 2;
        wg_per_cu = std::min(n, kStreamWgPerCu);
    }
    const unsigned blocks = (unsigned)std::min<long long>(((long long)sp.N + 3) / 4, (long long)n_cu * wg_per_cu);
    GNNCCA_LAUNCH((mpn_step_stream_kernel<FIRST, CLS, MSG, EB, NT>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool FIRST, bool CLS, bool MSG>
static hipError_t launch_stream(const StepParams& sp, hipStream_t st, int n_cu) {
    const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
    if (nt == 2) return sp.e_bf16 ? launch_stream_t<FIRST, CLS, MSG, true, 2>(sp, st, n_cu) : launch_stream_t<FIRST, CLS, MSG, false, 2>(sp, st, n_cu);
    if (nt == 1) return sp.e_bf16 ? launch_stream_t<FIRST, CLS, MSG, true, 1>(sp, st, n_cu) : launch_stream_t<FIRST, CLS, MSG, false, 1>(sp, st, n_cu);
    return sp.e_bf16 ? launch_stream_t<FIRST, CLS, MSG, true, 0>(sp, st, n_cu) : launch_stream_t<FIRST, CLS, MSG, false, 0>(sp, st, n_cu);
}

static hipError_t launch_stream_dispatch(const StepParams& sp, bool msg, hipStream_t st, int n_cu) {
    const int key = (sp.first ? 4 : 0) | (sp.cls_layers ? 2 : 0) | (msg ? 1 : 0);
    switch (key) {
        case 0: return launch_stream<false, false, false>(sp, st, n_cu);
        case 1: return launch_stream<false, false, true>(sp, st, n_cu);
        case 2: return launch_stream<false, true, false>(sp, st, n_cu);
        case 3: return launch_stream<false, true, true>(sp, st, n_cu);
        case 4: return launch_stream<true, false, false>(sp, st, n_cu);
        case 5: return launch_stream<true, false, true>(sp, st, n_cu);
        case 6: return launch_stream<true, true, false>(sp, st, n_cu);
        case 7: return launch_stream<true, true, true>(sp, st, n_cu);
    }
    return hipErrorInvalidValue;
}

}  // namespace gnncca
