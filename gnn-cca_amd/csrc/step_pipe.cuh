#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Step kernel of the shipped GRAPH_NET_PARAMS shape, round 3: the algorithm of mpn_step_kernel (step_general.cuh) with
//   * the node message on the bf16 matrix pipe in split form (msg_bf16.cuh) -- the f32-input MFMAs of rounds 1-2 turned
//     out to serialise with the VALU of the whole SIMD and were 40 % of the kernel's issue time;
//   * all streams through BUFFER instructions (raw buffer resource + 32-bit offsets): a lane beyond the segment gets
//     the out-of-range offset 2^31 -- its loads return 0 and its stores are dropped by the address unit: no exec mask,
//     no branch inside a round, and the feature-plane offset rides in an SGPR (no 64-bit VALU address arithmetic:
//     -25 VALU per chunk); the tail mask of the aggregation rides in a k-slot of the MFMA (msg_bf16.cuh);
//   * the body of a ROUND (two 64-edge chunks of a wave) as ONE basic block whose program order alternates one MFMA with
//     6-10 VALU instructions of another dependency chain (edge update / classifier of the other chunk, ReLU +
//     accumulate of a finished tile); three accumulator tiles rotate.  __builtin_amdgcn_sched_barrier keeps hipcc from
//     regrouping; the empty asm statements in `ra` pin pure arithmetic that instruction selection would otherwise sink.
// Work split, LDS tables, prefetch of the first two rounds, cross-wave combine, projection epilogue and cache policies
// are mpn_step_fast_kernel's of round 2 (step_fast.cuh, kept as the A/B reference behind GNNCCA_DIAG).
// ------------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;   // >= every buffer's num_records (all <= 2^31, checked on the host)

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned long long bytes) {
    // raw buffer (stride 0), 32-bit data format word of gfx9 (0x00020000), range-checked against `bytes`
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float buf_load_f32(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}

// host-side eligibility: every buffer this kernel addresses with 32-bit offsets stays below 2^31 bytes
static inline bool step_pipe_fits(long long N, long long E, long long e_stride, unsigned long long ws_bytes) {
    const long long lim = 1ll << 31;
    // (the whole workspace is addressed through one descriptor: it must end below the out-of-range offset 2^31 of a dead lane, and
    // below 2^31 - 4 so that "the slot before slot 0" of DERIVE's predecessor load is out of range too)
    return 6 * e_stride * 4 <= lim && E * 16 <= lim && N * 32 <= lim && ws_bytes <= (unsigned long long)lim - 4096;
}

#define GNNCCA_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef GNNCCA_NPW2_CLS
#define GNNCCA_NPW2_CLS 0         // 1: instantiate the two-nodes-per-wave form for the classifying steps as well (diagnostic builds)
#endif
#ifndef GNNCCA_NPW2_CLS_WAVES
#define GNNCCA_NPW2_CLS_WAVES 3   // waves per SIMD the CLASSIFYING two-nodes-per-wave variants are compiled for (143-155 VGPRs; the others fit four)
#endif

// RNG: the column-range code is compiled in (StepParams::rng) -- step 1 derives the ranges, later steps compute target ids from them.  A
// template parameter and not only a run-time switch because this kernel's scalar register file is full: the few SGPRs the ranges need
// turn into v_readlane / v_writelane spill traffic on the VALU, which is what the issue-bound big batches are short of; the host
// instantiates RNG for the latency-bound regime only (mpn_forward.hip).
// NPW = 2 (round 4): a wave owns TWO consecutive nodes (one wave per node only; message steps; no LDS table, no range code).  Batches of
// small dense graphs give a wave ONE round (<= 128 edges) per node: its whole life is the chain seg_ptr -> ids + state -> gather ->
// arithmetic -> epilogue with nothing to overlap but the other three waves of the SIMD, and 64 x dense256 (two rounds per wave, the second
// requested from inside the first) streams at 0.66 of HBM where 512 x dense128 reaches 0.57.  Here the second node's round takes the
// place of "the second round": its ids are requested before the first node is computed, its edge state from inside the first node's
// arithmetic (the hook), its gather before the first node's epilogue; weights, staging and the workgroup barrier are paid once for both.
// CIN (round 4): this step classifies its INPUT edge state -- the latents the previous step left in HBM -- into the previous step's logit
// slot (StepParams::logits_in).  Same classifier, same fp32 values, same bits as classifying them where they were produced; it moves the
// classifier out of the message steps that produce a classified state and into the step that reads it back anyway, so that those steps
// run the lighter variant (and its two-nodes-per-wave form).  fp32 edge state only: the bf16 state is rounded AFTER its step classified it.
template <bool FIRST, bool CLS, bool MSG, bool PD_LDS, bool EBF16, int NT, bool RNG = false, int NPW = 1, bool CIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MSG ? (NPW == 2 && CLS ? GNNCCA_NPW2_CLS_WAVES : 4) : 1))) void mpn_step_pipe_kernel(const StepParams p) {
    static_assert(NPW == 1 || (NPW == 2 && MSG && !PD_LDS && !RNG), "two nodes per wave: message steps without LDS table / range code");
    static_assert(!CIN || (!FIRST && !EBF16 && !RNG && !PD_LDS), "classifying the input state: later steps, fp32 edge state");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_part = s_proj + (MSG ? kH * kProjOut : 0);     // [4][32]
    int* s_rng = reinterpret_cast<int*>(s_part + 4 * kH);   // [4][4]     (DERIVE: per-wave column-range findings)
    float* s_pd = s_part + 4 * kH + 16;                     // [N + 1][8] (PD_LDS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);
    constexpr bool DERIVE = FIRST && MSG && RNG;   // step 1 of a forward with more steps to come: derive the column ranges (StepParams::rng)
    constexpr int AUX_ST = NT >= 1 ? 2 : 0;    // nt: streams written / read once (see mpn_step_fast_kernel on the policy)
    constexpr int AUX_LD = NT >= 2 ? 2 : 0;

    GNNCCA_STAMP(p.stamp_slot, 0);
    // ---- prologue: every independent load is issued before the first wait ----------------------------------
    const int ktouch = NPW == 1 ? touch_kernargs(sizeof(StepParams)) : 0;   // (common.cuh; the two-node forms have no SGPRs to spare)
    const unsigned gflags = p.flags[0];
    const unsigned rbad = p.flags[1];
    const int wps = NPW == 2 ? 1 : p.wps;   // 1, 2 or 4 (mpn_forward.hip): shifts, not a software division
    const int wl = wps >> 1;                // log2(wps)
    int node = NPW == 2 ? (blockIdx.x * 4 + wave) * 2 : (int)(blockIdx.x << (2 - wl)) + (wave >> wl);
    const int sub = NPW == 2 ? 0 : wave & (wps - 1);
    bool active = node < p.N;
    int nclamp = active ? node : 0;
    int seg_s = p.seg_ptr[nclamp];
    int seg_t = p.seg_ptr[nclamp + 1];
    // NPW == 2: the second node is node + 1; its segment starts where the first one's ends
    const bool active2 = NPW == 2 && node + 1 < p.N;
    int seg_t2 = NPW == 2 ? p.seg_ptr[active2 ? node + 2 : 0] : 0;
    // steps 2 ... L: the node's column ranges as step 1 left them (wave-uniform: four SGPRs)
    const bool use_range = RNG && !FIRST && rbad == 0u;
    int rbk = 0, rA = 0, rB = 0;
    if (RNG && !FIRST) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 r = reinterpret_cast<const i32x4*>(p.rng)[nclamp];
        rbk = seg_s + r[1], rA = r[0] - seg_s, rB = r[2] - seg_s;
    }
    // ONE scalar round trip for the flags, the CSR offsets and the column ranges (see mpn_step_fast_kernel)
    if (NPW == 1) asm volatile("" ::"s"(gflags), "s"(rbad), "s"(seg_s), "s"(seg_t), "s"(rbk), "s"(rA), "s"(rB), "s"(ktouch));
    const int ch = lane & 31;
    const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
    float psrc[kEF];
    MsgB mb;                // B operands of the message MFMAs (msg_bf16.cuh)
    float cinit = 0.f;
    float cinit2 = 0.f;     // NPW == 2: Q[node + 1][ch]
    f32x4 stage_proj[2];
    f32x4 stage_pd[8];
    float projb_l = 0.f;
    const int pd_n4 = p.N * (kPdStride / 4);
    // requested AFTER the first round's target ids (see mpn_step_fast_kernel: loads return in order, the ids are waited for first)
    auto request_node_operands = [&]() {
#pragma unroll
    for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
    if (NPW == 2) cinit2 = p.psq_in[(size_t)(active2 ? node + 1 : 0) * kPsQStride + 8 + ch];
    if (MSG) {
        cinit = psq[8 + ch];
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
        msg_b_weights(blob + p.off_wnebf, lane, mb);
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        // 384 float4 in all.  The second request and its store are UNCONDITIONAL (threads 128 ... 255 repeat their first float4): under
        // `if (tid + 256 < 384)` the compiler sinks the load into the branch, next to its store, behind s_waitcnt vmcnt(0) -- a round trip
        // for every load of the prologue, in waves 0-1, which the other waves then sit out at the barrier
        stage_proj[0] = g4[tid];
        stage_proj[1] = g4[tid + 256 < kH * kProjOut / 4 ? tid + 256 : tid];
    }
    if (PD_LDS) {  // N <= 1024: at most 8 float4 per thread
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.pd_in);
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_pd[i] = g4[min(tid + i * 256, pd_n4 - 1)];
    }
    };
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS || CIN)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256) {
                if (CLS) p.logits[k] = __builtin_nanf("");
                if (CIN) p.logits_in[k] = __builtin_nanf("");
            }
        return;
    }
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    if (!active) seg_s = seg_t = 0;
    if (NPW == 2 && !active2) seg_t2 = seg_t;   // (no second node: an empty segment)
    // padded layout: see mpn_step_fast_kernel
    const bool padded = p.ell_S > 0 && !(gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR));
    int eoff = padded ? nclamp * p.ell_S - seg_s : 0;
    const int eoff2 = (NPW == 2 && padded) ? (node + 1) * p.ell_S - seg_t : 0;   // the second node's
    const unsigned plane_b = (unsigned)p.e_stride * 4u;                       // bytes between two feature planes
    const unsigned long long live = (p.diag & 1) ? 0ull : 1ull;               // timing-only diagnostic: every stream descriptor empty
    // ONE descriptor for everything this kernel streams out of the forward's workspace -- edge state, target ids, permutation, P_dst
    // table -- with the region's byte offset as the scalar offset of each access (round 4; rounds 1-3 kept a descriptor per buffer:
    // 16 SGPRs where 4 + 3 do, in a kernel whose scalar file is full -- hipcc spills SGPRs into VGPR lanes with v_writelane /
    // v_readlane, i.e. on the VALU, the unit this kernel is bound by: 23 of the 430 instructions of a steady-state round).  The host
    // guarantees ws_bytes < 2^31 - 4096 (step_pipe_fits), so the out-of-range offset 2^31 of a dead lane is beyond it for every region.
    const rsrc_t r_ws = make_rsrc(p.ws_base, live * p.ws_bytes);
    const unsigned so_e = p.so_e, so_col = p.so_col, so_perm = p.so_perm, so_pd = p.so_pd;
    const unsigned us_oob = unsorted ? 0u : kOobOffset;                       // sorted rows: the permutation is never fetched (its loads return 0)
    const rsrc_t r_attr = make_rsrc(p.edge_attr, live * (unsigned long long)p.E * 16);
    // one descriptor for the logits: with CIN the previous step's slot, and -- when this step classifies its output as well (the last step) --
    // its own slot right behind it (the host hands out consecutive slots of the [n_out][E] buffer)
    const rsrc_t r_log = make_rsrc(CIN ? p.logits_in : p.logits, (CLS || CIN) ? live * (unsigned long long)p.E * (CLS && CIN ? 8 : 4) : 0ull);
    const unsigned so_log_out = (CLS && CIN) ? (unsigned)p.E * 4u : 0u;

    f32x16 acc;  // 'sum' / 'mean' only: 'max' takes the general kernel
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

#include "step_pieces.inc"
    const int stride = 64 * wps;
    int base = seg_s + 64 * sub;
    Chunk c0, c1;
    // (register budget: the two-node forms sit at 126-128 VGPRs and spill with the ids requested first -- they keep the order of round 4)
    constexpr bool IDS_FIRST = NPW == 1;
    if (!IDS_FIRST) {
        request_node_operands();
        if (MSG) msg_b_bias(cinit, lane, mb);
    }
    load_index(base, c0);
    load_index(base + stride, c1);
    if (IDS_FIRST) {
        GNNCCA_SB();   // (the scheduler would cluster these loads in its own order)
        request_node_operands();
        GNNCCA_SB();
    }
    load_state(base, c0);
    load_state(base + stride, c1);
    if (IDS_FIRST) GNNCCA_SB();   // (and would put the gathers' address arithmetic -- a wait for the ids -- ahead of the state requests)
    // the first round's P_dst gathers go out before the staging stores and the barrier (not when they read the LDS table): the barrier's
    // skew runs under their round trip instead of ahead of it
    // (register budget: the two-node forms and step 1 classifying sit at 126-128 VGPRs and would spill -- they keep the old order)
    constexpr bool PRE_GATHER = !PD_LDS && NPW == 1 && !(FIRST && CLS);
    if (PRE_GATHER) {
        load_target(c0);
        load_target(c1);
    }
    if (MSG) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        l4[tid] = stage_proj[0];
        l4[tid + 256 < kH * kProjOut / 4 ? tid + 256 : tid] = stage_proj[1];
    }
    if (PD_LDS) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_pd);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (tid + i * 256 < pd_n4) l4[tid + i * 256] = stage_pd[i];
        if (tid < kPdStride / 4) l4[pd_n4 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};   // row N: what a lane beyond its segment gathers
    }
    if (MSG && IDS_FIRST) msg_b_bias(cinit, lane, mb);
    GNNCCA_STAMP(p.stamp_slot, 1);
    // INVARIANT (shared with mpn_step_fast_kernel): between the staging stores above and the combine's __syncthreads() below NOTHING reads
    // s_proj or s_part, and no wave returns or skips that barrier (the BAD_INDEX return above is block-uniform and precedes the stores).
    // With several waves per node the cross-wave combine's barrier is therefore the one that publishes s_proj to the epilogue, its only
    // reader, and no early barrier is needed unless the gathers read s_pd.  A new LDS read in between, or a per-wave early exit, turns
    // this into a silent race: GNNCCA_STEP_EARLYBAR (diag bit 3) restores the early barrier to bisect such a change.
    if (PD_LDS || (MSG && (wps == 1 || (p.diag & 8)))) __syncthreads();   // (diag bit 3: A/B with the early barrier of rounds 1-2)
    GNNCCA_STAMP(p.stamp_slot, 2);
    auto round_compute = [&](int rb, Chunk& a, Chunk& b) {
        if (rb + stride < seg_t) {
            compute2(rb, a, rb + stride, b);
        } else {
            if (NPW == 2) hook_fire();   // (compute2 fires the hook from inside its arithmetic; a lone chunk fires it up front)
            compute1(rb, a);
        }
    };
    auto round_body = [&](int rb, Chunk& a, Chunk& b, int sid, int sland, int sdone, bool requested = false) {
        if (!requested) {
            load_target(a);
            load_target(b);
        }
        if (DERIVE) {
            derive(rb, a);
            if (rb + stride < seg_t) derive(rb + stride, b);
        }
        GNNCCA_STAMP(p.stamp_slot, sid);
#ifdef GNNCCA_STAMPS   // diagnostic build: when did the round's operands arrive?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GNNCCA_STAMP(p.stamp_slot, sland);
#endif
        if (!MSG) hook_fire();   // the message-less last step has registers to spare: its second round's state goes out right away
        round_compute(rb, a, b);
        GNNCCA_STAMP(p.stamp_slot, sdone);
    };
    auto finish_node = [&]() {
        float v = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += acc[i];
        v += __shfl_xor(v, 32);
        if (wps > 1) {
            if (lane < kH) s_part[wave * kH + lane] = v;
            if (DERIVE && lane == 0) s_rng[wave * 4] = d_nb, s_rng[wave * 4 + 1] = d_first, s_rng[wave * 4 + 2] = d_start2;
            __syncthreads();
            if (sub == 0) {
                v = s_part[wave * kH + ch];
                for (int u = 1; u < wps; ++u) v += s_part[(wave + u) * kH + ch];
                if (DERIVE)
                    for (int u = 1; u < wps; ++u) {
                        d_nb += s_rng[(wave + u) * 4];
                        const int f_u = s_rng[(wave + u) * 4 + 1];
                        if (f_u < d_first) d_first = f_u, d_start2 = s_rng[(wave + u) * 4 + 2];
                    }
            }
        }
        if (DERIVE && active && sub == 0 && lane == 0) {   // (start1, len1, start2 - len1, breaks) of this node
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const int len1 = d_nb ? d_first : seg_t - seg_s;
            reinterpret_cast<i32x4*>(p.rng)[node] = i32x4{d_start1, len1, d_start2 - len1, d_nb};
            if (d_nb > 1) atomicOr(p.flags + 1, 1u);   // not two runs: every later step of this forward streams col32
        }
        GNNCCA_STAMP(p.stamp_slot, 6);
        if (active && sub == 0 && !(p.diag & 2)) {   // (diag bit 1: timing-only run without the projection epilogue)
            const int deg = seg_t - seg_s;
            if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
            if (deg == 0) v = 0.f;
            // projection epilogue: project_node's ordered FMA chain (same bits), with h' broadcast through 128 B of LDS -- one
            // ds_write + eight broadcast ds_read_b128 on the LDS pipe instead of 32 v_readlane on the VALU, which is the unit this
            // kernel is short of (the row of s_part is this wave's own; LDS operations of a wave execute in order)
            const int o = min(lane, kProjOut - 1);
            float pr = projb_l;
            const float* w = s_proj + o;
            float* hrow = s_part + wave * kH;
            if (lane < kH) hrow[lane] = v;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c4 = 0; c4 < kH / 4; ++c4) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + 4 * c4);
#pragma unroll
                for (int q = 0; q < 4; ++q) pr = fmaf(w[(4 * c4 + q) * kProjOut], hv[q], pr);
            }
            if (lane < kPdStride)
                p.pd_out[(size_t)node * kPdStride + lane] = pr;
            else if (lane < kProjOut)
                p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
        }
    };
    // NPW == 2: the wave's second node takes the first one's place -- same arithmetic, same order per node (logits and latents do not
    // depend on NPW).  (The epilogue's LDS row of this wave is written again only after it was read: a wave's LDS operations execute in order.)
    bool second = false;
    auto switch_node = [&]() {
        second = true;
        node = node + 1, active = active2;
        seg_s = seg_t, seg_t = seg_t2, eoff = eoff2;
        const float* __restrict__ psq2 = p.psq_in + (size_t)(active ? node : 0) * kPsQStride;
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = psq2[f];
        msg_b_bias(cinit2, lane, mb);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    };
    if (base < seg_t) {
        // the SECOND round's target ids are requested before the first round is computed (see mpn_step_fast_kernel)
        // The second round's edge state is requested from INSIDE the first round's arithmetic (hook_fire in compute2, right after the
        // edge updates, when the first pair's loaded operands are dead): its HBM round trip runs under the message block instead
        // of after it.  (First attempt, with three accumulator tiles: 133 registers, 46-47 -> 49-52 us, r3_ab_hook1.log; with two
        // tiles it fits 127 registers: -1...-5 % on the message steps of 64 x dense256 / 200 x dense256, r3_ab_hook3.log.)
        const int base2 = base + 2 * stride;
        Chunk& n0 = hook_a;
        Chunk& n1 = hook_b;
        hook_stride = stride;
        if (NPW == 2 && base2 >= seg_t && seg_t < seg_t2) {
            // ONE round for this node: the hooked round is the NEXT node's first (its segment starts at seg_t).  The whole hand-over sits in
            // this branch so that the hooked chunks are live across nothing else (kept live across the loop below they cost 19 registers)
            load_index_at(seg_t, n0, seg_t2);
            load_index_at(seg_t + stride, n1, seg_t2);
            // (the variant that classifies its input has no registers for the second node's edge state under the first node's arithmetic --
            // 36 B of scratch at four waves per SIMD, and a kernel with ANY scratch starts its waves far more slowly; it requests the state
            // after the hand-over, with the gather: one exposed round trip instead of three, 127 VGPRs)
            if (!CIN) hook_base = seg_t, hook_end = seg_t2, hook_eoff = eoff2;
            round_body(base, c0, c1, 3, 15, 4, PRE_GATHER);
            // the second node's gather: before the first node's epilogue where the registers allow it (12 more across the epilogue)
            if (!CLS && !CIN) {
                load_target(n0);
                load_target(n1);
            }
            finish_node();
            switch_node();
            if (CIN) {
                load_state(seg_s, n0);
                load_state(seg_s + stride, n1);
            }
            if (CLS || CIN) {
                load_target(n0);
                load_target(n1);
            }
            round_compute(seg_s, n0, n1);
            base = seg_s + 2 * stride;
        } else {
            const bool use_hook = !PD_LDS && base2 < seg_t && base + stride < seg_t && !(p.diag & 4);   // (diag bit 2: A/B without it)
            if (!PD_LDS) {
                load_index(base2, n0);
                load_index(base2 + stride, n1);
                if (use_hook) hook_base = base2, hook_end = seg_t, hook_eoff = eoff;
            }
            round_body(base, c0, c1, 3, 15, 4, PRE_GATHER);
            if (base2 < seg_t) {
                if (PD_LDS) {
                    load_index(base2, n0);
                    load_index(base2 + stride, n1);
                }
                if (!use_hook) {
                    load_state(base2, n0);
                    load_state(base2 + stride, n1);
                }
                round_body(base2, n0, n1, 8, 10, 9);
            }
            base += 4 * stride;
        }
    }
    for (; base < seg_t; base += 2 * stride) {
        load_index(base, c0);
        load_index(base + stride, c1);
        load_state(base, c0);
        load_state(base + stride, c1);
        round_body(base, c0, c1, 11, 13, 12);
    }
    GNNCCA_STAMP(p.stamp_slot, 5);
    if (MSG) finish_node();   // (the first node's, or -- after the hand-over above -- the second one's)
    if (NPW == 2 && !second) {   // the second node from a cold start (the first one had no edges or several rounds)
        switch_node();
        for (base = seg_s; base < seg_t; base += 2 * stride) {
            load_index(base, c0);
            load_index(base + stride, c1);
            load_state(base, c0);
            load_state(base + stride, c1);
            round_body(base, c0, c1, 11, 13, 12);
        }
        finish_node();
    }
    GNNCCA_STAMP(p.stamp_slot, 7);
}

template <bool FIRST, bool CLS, bool MSG, bool PDL, bool EB, int NT, bool RNG = false, int NPW = 1, bool CIN = false>
static hipError_t launch_pipe_t(const StepParams& sp, hipStream_t st) {
    const int npg = NPW == 2 ? 8 : 4 / sp.wps;
    const unsigned blocks = (unsigned)((sp.N + npg - 1) / npg);
    const size_t lds = ((MSG ? (size_t)kH * kProjOut : 0) + 4 * kH + 16 + (PDL ? ((size_t)sp.N + 1) * kPdStride : 0)) * sizeof(float);
    GNNCCA_LAUNCH((mpn_step_pipe_kernel<FIRST, CLS, MSG, PDL, EB, NT, RNG, NPW, CIN>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

// A step of a forward with DEFERRED classification (forward_impl: StepParams::logits_in != nullptr) that classifies its input state:
// a message step (never classifies its own output in that scheme) or the last step (classifies both).  fp32 state, no LDS table, no ranges.
static hipError_t launch_pipe_cin(const StepParams& sp, bool msg, hipStream_t st) {
    const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
    if (msg) {
        if (sp.npw == 2 && sp.wps == 1) {   // two nodes per wave; the second node's ids early, its state and gather after the hand-over (registers)
            if (nt == 2) return launch_pipe_t<false, false, true, false, false, 2, false, 2, true>(sp, st);
            if (nt == 1) return launch_pipe_t<false, false, true, false, false, 1, false, 2, true>(sp, st);
            return launch_pipe_t<false, false, true, false, false, 0, false, 2, true>(sp, st);
        }
        if (nt == 2) return launch_pipe_t<false, false, true, false, false, 2, false, 1, true>(sp, st);
        if (nt == 1) return launch_pipe_t<false, false, true, false, false, 1, false, 1, true>(sp, st);
        return launch_pipe_t<false, false, true, false, false, 0, false, 1, true>(sp, st);
    }
    if (nt == 2) return launch_pipe_t<false, true, false, false, false, 2, false, 1, true>(sp, st);
    if (nt == 1) return launch_pipe_t<false, true, false, false, false, 1, false, 1, true>(sp, st);
    return launch_pipe_t<false, true, false, false, false, 0, false, 1, true>(sp, st);
}

// two nodes per wave (StepParams::npw == 2: the host's choice, mpn_forward.hip): message steps, one wave per node, no LDS table, no range code.
// Only the steps that do not classify are instantiated: their two-node form fits four waves per SIMD (115-128 VGPRs) and gains 11-16 % on
// batches of >= 32 768 nodes (512 x dense128 step 1: 117.7 -> 105.2 us; 1000 x dense64: 88.6 -> 74.3); the classifying variants need 143-155
// registers, i.e. three waves per SIMD (or scratch: 2x slower), and come out at -7 ... +3 % (profiles/r04_logs/ab_npw{1,2,3,4}.log).
template <bool FIRST, bool CLS, bool MSG, bool PDL, bool EB, int NT>
static hipError_t launch_pipe_npw(const StepParams& sp, hipStream_t st) {
    if constexpr (MSG && !PDL && (!CLS || GNNCCA_NPW2_CLS)) {
        if (sp.npw == 2 && sp.wps == 1 && sp.rng == nullptr) return launch_pipe_t<FIRST, CLS, MSG, PDL, EB, NT, false, 2>(sp, st);
    }
    return launch_pipe_t<FIRST, CLS, MSG, PDL, EB, NT>(sp, st);
}

template <bool FIRST, bool CLS, bool MSG, bool PDL>
static hipError_t launch_pipe(const StepParams& sp, hipStream_t st) {
    if (!PDL) {   // the non-temporal variants only exist beyond the LDS-resident gather table (N > 1024): big batches
        const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
        if (nt == 2) return sp.e_bf16 ? launch_pipe_npw<FIRST, CLS, MSG, false, true, 2>(sp, st) : launch_pipe_npw<FIRST, CLS, MSG, false, false, 2>(sp, st);
        if (nt == 1) return sp.e_bf16 ? launch_pipe_npw<FIRST, CLS, MSG, false, true, 1>(sp, st) : launch_pipe_npw<FIRST, CLS, MSG, false, false, 1>(sp, st);
    }
    // the column-range variants (StepParams::rng != nullptr: the host's choice) exist for the default cache policy and where they do
    // something: step 1 when it derives (FIRST && MSG), every later step
    if (sp.rng != nullptr && (!FIRST || MSG))
        return sp.e_bf16 ? launch_pipe_t<FIRST, CLS, MSG, PDL, true, 0, true>(sp, st) : launch_pipe_t<FIRST, CLS, MSG, PDL, false, 0, true>(sp, st);
    return sp.e_bf16 ? launch_pipe_npw<FIRST, CLS, MSG, PDL, true, 0>(sp, st) : launch_pipe_npw<FIRST, CLS, MSG, PDL, false, 0>(sp, st);
}

#ifndef GNNCCA_KERNELS_ONLY   // (tools: compile-only probes of single instantiations skip the dispatch tables)
static hipError_t launch_pipe_dispatch(const StepParams& sp, bool msg, hipStream_t st) {
    if (sp.logits_in != nullptr) return launch_pipe_cin(sp, msg, st);
    const int key = (sp.first ? 8 : 0) | (sp.cls_layers ? 4 : 0) | (msg ? 2 : 0) | (sp.pd_lds ? 1 : 0);
    switch (key) {
#define GNNCCA_PIPE_CASE(K, A, B, C, D) \
    case K: return launch_pipe<A, B, C, D>(sp, st);
        GNNCCA_PIPE_CASE(0, false, false, false, false)
        GNNCCA_PIPE_CASE(1, false, false, false, true)
        GNNCCA_PIPE_CASE(2, false, false, true, false)
        GNNCCA_PIPE_CASE(3, false, false, true, true)
        GNNCCA_PIPE_CASE(4, false, true, false, false)
        GNNCCA_PIPE_CASE(5, false, true, false, true)
        GNNCCA_PIPE_CASE(6, false, true, true, false)
        GNNCCA_PIPE_CASE(7, false, true, true, true)
        GNNCCA_PIPE_CASE(8, true, false, false, false)
        GNNCCA_PIPE_CASE(9, true, false, false, true)
        GNNCCA_PIPE_CASE(10, true, false, true, false)
        GNNCCA_PIPE_CASE(11, true, false, true, true)
        GNNCCA_PIPE_CASE(12, true, true, false, false)
        GNNCCA_PIPE_CASE(13, true, true, false, true)
        GNNCCA_PIPE_CASE(14, true, true, true, false)
        GNNCCA_PIPE_CASE(15, true, true, true, true)
#undef GNNCCA_PIPE_CASE
    }
    return hipErrorInvalidValue;
}
#endif

}  // namespace gnncca
