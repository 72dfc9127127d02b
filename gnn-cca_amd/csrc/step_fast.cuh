#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Specialised step kernel for the shipped GRAPH_NET_PARAMS shape (edge_in 4, no reattach, classifier 6->4->1 or
// off, no debug taps): the same algorithm as mpn_step_kernel with every per-config decision made at compile
// time, so the body is straight-line code whose loads issue back to back.  mpn_step_kernel stays as the
// general / traced variant.
// ------------------------------------------------------------------------------------------------------------
// ReLU of an MFMA result as a signed-integer max: negative floats (sign bit set) are negative integers, -0 included.  One
// VALU op; fmaxf first canonicalises the (not provably canonical) MFMA output with a second v_max per element -- 32 extra
// VALU instructions per 64-edge chunk.  NaN inputs: a positive NaN stays, a negative one becomes 0 (fmaxf gives 0 for both).
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

#ifdef GNNCCA_FAST_WAVES   // diagnostic builds: force the register budget of N waves per SIMD
#define GNNCCA_FAST_ATTR __attribute__((amdgpu_waves_per_eu(GNNCCA_FAST_WAVES, GNNCCA_FAST_WAVES)))
#else
#define GNNCCA_FAST_ATTR __attribute__((amdgpu_waves_per_eu(MSG ? 4 : 1)))   // the message variants must fit four waves per SIMD (128 VGPRs); the others stream: keep them light
#endif
template <bool FIRST, bool CLS, bool MSG, bool PD_LDS, bool EBF16, int NT>
__global__ __launch_bounds__(256) GNNCCA_FAST_ATTR void mpn_step_fast_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_part = s_proj + (MSG ? kH * kProjOut : 0);     // [4][32]
    int* s_rng = reinterpret_cast<int*>(s_part + 4 * kH);   // [4][4]     (DERIVE: per-wave column-range findings)
    float* s_pd = s_part + 4 * kH + 16;                     // [N][8]     (PD_LDS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    constexpr bool DERIVE = FIRST && MSG;      // step 1 of a forward with more steps to come: derive the column ranges (StepParams::rng)
    // Per-step scalars (152 floats) are read through the CONSTANT address space: wave-uniform addresses there
    // become s_load into SGPRs, which the VALU takes as operands directly -- no LDS, no VGPR copies.
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);

    GNNCCA_STAMP(p.stamp_slot, 0);
    // ---- prologue: every independent load is issued before the first wait ----------------------------------
    const int ktouch = touch_kernargs(sizeof(StepParams));
    const unsigned gflags = p.flags[0];
    const unsigned rbad = p.flags[1];
    const int wps = p.wps;         // 1, 2 or 4 (mpn_forward.hip): shifts, not the 40-instruction software division of round 1-4
    const int wl = wps >> 1;       // log2(wps)
    const int node = (int)(blockIdx.x << (2 - wl)) + (wave >> wl);
    const int sub = wave & (wps - 1);
    const bool active = node < p.N;
    const int nclamp = active ? node : 0;
    int seg_s = p.seg_ptr[nclamp];
    int seg_t = p.seg_ptr[nclamp + 1];
    // steps 2 ... L: the node's column ranges as step 1 left them (StepParams::rng; wave-uniform: four SGPRs)
    const bool use_range = !FIRST && p.rng != nullptr && rbad == 0u;
    int rs1 = 0, rl1 = 0, rd2 = 0;
    if (!FIRST && p.rng != nullptr) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 r = reinterpret_cast<const i32x4*>(p.rng)[nclamp];
        rs1 = r[0], rl1 = r[1], rd2 = r[2];
    }
    // ONE scalar round trip for the flags, the CSR offsets and the column ranges: without this pin the compiler hoists the BAD_INDEX test
    // (a wait for the flags alone) above the other scalar loads and the launch starts with two dependent L2 round trips instead of one.
    asm volatile("" ::"s"(gflags), "s"(rbad), "s"(seg_s), "s"(seg_t), "s"(rs1), "s"(rl1), "s"(rd2), "s"(ktouch));
    const int nmax = p.N - 1;
    const int half = lane >> 5, ch = lane & 31;
    const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
    float psrc[kEF];
    float cinit = 0.f;
    float bw[3] = {0.f, 0.f, 0.f};
    f32x4 stage_proj[2];
    f32x4 stage_pd[8];
    float projb_l = 0.f;
    const int pd_n4 = p.N * (kPdStride / 4);
    // The node's own operands, the weights and the tables staged through LDS.  Requested AFTER the first round's target ids (below): a
    // wave's loads return in order, so what is waited for first -- the ids, which the P_dst gather's addresses need -- is asked for first.
    auto request_node_operands = [&]() {
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
        if (MSG) {
            cinit = psq[8 + ch];
            projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
#pragma unroll
            for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
            const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
            stage_proj[0] = g4[tid];                                   // 384 float4 in all
            stage_proj[1] = g4[min(tid + 256, kH * kProjOut / 4 - 1)];
        }
        if (PD_LDS) {  // N <= 1024: at most 8 float4 per thread
            const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.pd_in);
#pragma unroll
            for (int i = 0; i < 8; ++i) stage_pd[i] = g4[min(tid + i * 256, pd_n4 - 1)];
        }
    };
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    if (!active) seg_s = seg_t = 0;
    // Edge-state slot of sorted edge kk: kk + eoff.  Padded layout (big, nearly regular batches): the node's segment starts at
    // node * ell_S, a multiple of 128 B in every feature plane, so no line of the state is shared by two segments (with the
    // compact order a wave's 256-B accesses straddle three lines and every boundary line crosses the fabric twice: 12-27 %
    // more bytes than the algorithm needs).  The plan has checked every degree against ell_S; if one did not fit
    // (GNNCCA_GRAPH_IRREGULAR) or the rows were not sorted, every step of this forward uses the compact order.
    const long long eoff = (p.ell_S > 0 && !(gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR)))
                               ? (long long)nclamp * p.ell_S - seg_s : 0ll;   // (inactive waves: slot 0, loads only)

    // Cache policy of the streams (template NT: 0 none, 1 stores + edge_attr, 2 + loads of e; chosen by the host from the size of the
    // edge state; a RUN-TIME flag does not work: the compiler merges the two arms of the branch into one plain access): while the
    // state of a step fits the 256 MB Infinity Cache next to everything else the step touches, the next step finds it there and
    // default-policy accesses are best (64 x dense256, 100 MB: non-temporal loads cost +20 %).  Beyond that the streams only evict
    // each other: non-temporal stores of e' / logits and loads of edge_attr -7 % at 512 x dense128 (200 MB of state), non-temporal
    // loads of e on top -6 % at 200 x dense256 (313 MB).
    constexpr bool nt_store = NT >= 1, nt_load = NT >= 2;
    f32x16 acc;  // 'sum' / 'mean' only: 'max' takes the general kernel
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int last = seg_t - 1;

    struct Chunk {
        float raw[kEF];
        float pd[kEF];
        int ko, j;
        int jp;   // DERIVE: the target id of the previous edge of the sorted order
    };
    // phase A: everything addressed by the edge slot itself (target id, permutation, edge state)
    auto load_index = [&](int base, Chunk& c) {
        const int kk = max(min(base + lane, last), 0);   // (an empty segment: slot 0, never used)
        c.ko = unsorted ? p.perm[kk] : kk;
        // kernel-uniform choice.  use_range: the node's columns are <= 2 contiguous runs -- the id is computed, no load, no round trip.
        // The arithmetic comes FIRST and unconditionally, the load overwrites it: written as if / else the compiler puts the load arm first
        // and, the two arms sharing the destination register, guards the computed arm with s_waitcnt vmcnt(0) -- a full round trip for
        // every load of the prologue before the edge state is even requested (rounds 3-5 until this was seen in the listing).
        int cj = 0;
        if (!FIRST) {
            const int q = kk - seg_s;
            cj = max(min(q + (q < rl1 ? rs1 : rd2), nmax), 0);   // (clamped: lanes of an empty segment gather a real row, never used)
            asm volatile("" : "+v"(cj));                           // (pins the arithmetic above the branch)
        }
        if (!use_range) {
            cj = p.col32[kk];
            if (DERIVE) c.jp = p.col32[max(kk - 1, 0)];   // same lines: an L1 hit
        }
        c.j = cj;
    };
    // Step 1 derives each node's column ranges from the target ids it has loaded anyway: a break is an edge whose target is not its
    // predecessor's + 1.  Wave-uniform bookkeeping on the scalar unit.
    int d_nb = 0, d_first = 0x7FFFFFFF, d_start2 = 0, d_start1 = 0;
    auto derive = [&](int base, const Chunk& c) {
        const int k = base + lane;
        const bool brk = k < seg_t && k > seg_s && c.j != c.jp + 1;
        const unsigned long long m = __ballot(brk);
        if (m != 0ull) {
            if (d_nb == 0) {   // chunks of a wave come in ascending order: the first break seen is the wave's first
                const int l = __ffsll((long long)m) - 1;
                d_first = base + l - seg_s;
                d_start2 = __builtin_amdgcn_readlane(c.j, l);
            }
            d_nb += __popcll(m);
        }
        if (base == seg_s) d_start1 = __builtin_amdgcn_readfirstlane(c.j);
    };
    auto load_state = [&](int base, Chunk& c) {
        const int kk = max(min(base + lane, last), 0);
        if (FIRST) {
            const f32x4* __restrict__ ap = reinterpret_cast<const f32x4*>(p.edge_attr + (size_t)c.ko * 4);
            const f32x4 a = nt_store ? __builtin_nontemporal_load(ap) : *ap;   // read once per forward
            c.raw[0] = a[0], c.raw[1] = a[1], c.raw[2] = a[2], c.raw[3] = a[3], c.raw[4] = 0.f, c.raw[5] = 0.f;
        } else if (EBF16) {
            // edge state stored as three planes of packed bf16 pairs: one dword load = two features
            const unsigned* __restrict__ e2 = reinterpret_cast<const unsigned*>(p.e);
#pragma unroll
            for (int f = 0; f < kEF / 2; ++f) {
                const unsigned w = nt_load ? __builtin_nontemporal_load(e2 + (size_t)f * p.e_stride + kk + eoff)
                                           : e2[(size_t)f * p.e_stride + kk + eoff];
                c.raw[2 * f] = __uint_as_float(w << 16);
                c.raw[2 * f + 1] = __uint_as_float(w & 0xFFFF0000u);
            }
        } else {
            if (nt_load) {
#pragma unroll
                for (int f = 0; f < kEF; ++f) c.raw[f] = __builtin_nontemporal_load(p.e + (size_t)f * p.e_stride + kk + eoff);
            } else {
#pragma unroll
                for (int f = 0; f < kEF; ++f) c.raw[f] = p.e[(size_t)f * p.e_stride + kk + eoff];
            }
        }
    };
    // phase B: the gather that depends on the target id
    auto load_target = [&](Chunk& c) {
        f32x4 a;
        f32x2 b2;
        if (PD_LDS) {
            a = *reinterpret_cast<const f32x4*>(s_pd + c.j * kPdStride);
            b2 = *reinterpret_cast<const f32x2*>(s_pd + c.j * kPdStride + 4);
        } else {
            const float* __restrict__ pdj = p.pd_in + (size_t)c.j * kPdStride;
            a = *reinterpret_cast<const f32x4*>(pdj);
            b2 = *reinterpret_cast<const f32x2*>(pdj + 4);
        }
        c.pd[0] = a[0], c.pd[1] = a[1], c.pd[2] = a[2], c.pd[3] = a[3], c.pd[4] = b2[0], c.pd[5] = b2[1];
    };
    auto compute_chunk = [&](int base, const Chunk& c) {
        const int k = base + lane;
        const bool valid = k < seg_t;
        float ein[kEF];
        if (FIRST) {
#pragma unroll
            for (int h = 0; h < kEF / 2; ++h) {
                f32x2 s = {cw[kFcEncB + 2 * h], cw[kFcEncB + 2 * h + 1]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 w = {cw[kFcEncW + q * kEF + 2 * h], cw[kFcEncW + q * kEF + 2 * h + 1]};
                    s = __builtin_elementwise_fma(w, f32x2{c.raw[q], c.raw[q]}, s);
                }
                ein[2 * h] = fmaxf(s[0], 0.f), ein[2 * h + 1] = fmaxf(s[1], 0.f);
            }
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) ein[f] = c.raw[f];
        }
        // packed fp32 (v_pk_fma_f32): two output features per instruction, weights as SGPR pairs from the
        // transposed [g][f] copy of W_ee
        float en[kEF];
        f32x2 s2[kEF / 2];
#pragma unroll
        for (int h = 0; h < kEF / 2; ++h) s2[h] = f32x2{psrc[2 * h], psrc[2 * h + 1]} + f32x2{c.pd[2 * h], c.pd[2 * h + 1]};
#pragma unroll
        for (int g = 0; g < kEF; ++g) {
            const f32x2 x = {ein[g], ein[g]};
#pragma unroll
            for (int h = 0; h < kEF / 2; ++h) {
                const f32x2 w = {cw[kFcWee + g * kEF + 2 * h], cw[kFcWee + g * kEF + 2 * h + 1]};
                s2[h] = __builtin_elementwise_fma(w, x, s2[h]);
            }
        }
#pragma unroll
        for (int h = 0; h < kEF / 2; ++h) en[2 * h] = fmaxf(s2[h][0], 0.f), en[2 * h + 1] = fmaxf(s2[h][1], 0.f);
        if (p.store_e && valid) {
            if (EBF16) {
                unsigned* __restrict__ e2 = reinterpret_cast<unsigned*>(p.e);
#pragma unroll
                for (int f = 0; f < kEF / 2; ++f) {
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    bf16x2_t pk;  // round to nearest even (v_cvt_pk_bf16_f32)
                    pk[0] = (__bf16)en[2 * f];
                    pk[1] = (__bf16)en[2 * f + 1];
                    if (nt_store)
                        __builtin_nontemporal_store(__builtin_bit_cast(unsigned, pk), e2 + (size_t)f * p.e_stride + k + eoff);
                    else
                        e2[(size_t)f * p.e_stride + k + eoff] = __builtin_bit_cast(unsigned, pk);
                }
            } else {
#pragma unroll
                for (int f = 0; f < kEF; ++f) {
                    if (nt_store)
                        __builtin_nontemporal_store(en[f], p.e + (size_t)f * p.e_stride + k + eoff);
                    else
                        p.e[(size_t)f * p.e_stride + k + eoff] = en[f];
                }
            }
        }
        if (CLS) {
            f32x2 z[2] = {f32x2{cw[kFcCb1], cw[kFcCb1 + 1]}, f32x2{cw[kFcCb1 + 2], cw[kFcCb1 + 3]}};
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                const f32x2 x = {en[f], en[f]};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 w = {cw[kFcCw1 + f * 4 + 2 * h], cw[kFcCw1 + f * 4 + 2 * h + 1]};
                    z[h] = __builtin_elementwise_fma(w, x, z[h]);
                }
            }
            float logit = cw[kFcCb2];
#pragma unroll
            for (int q = 0; q < 4; ++q) logit = fmaf(cw[kFcCw2 + q], fmaxf(z[q >> 1][q & 1], 0.f), logit);
            if (valid) {
                if (nt_store)
                    __builtin_nontemporal_store(logit, p.logits + c.ko);   // written once, read by the caller much later
                else
                    p.logits[c.ko] = logit;
            }
        }
        if (MSG) {
            f32x16 d0, d1;
#pragma unroll
            for (int i = 0; i < 16; ++i) d0[i] = d1[i] = cinit;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(en[2 * s]), __float_as_uint(en[2 * s + 1]),
                                                                false, false);
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[0]), bw[s], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[1]), bw[s], d1, 0, 0, 0);
            }
            if (base + 64 <= seg_t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = (acc[i] + relu_bits(d0[i])) + relu_bits(d1[i]);
            } else {
                const int rem = seg_t - base - 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int eo = (i & 3) + 8 * (i >> 2);
                    const float m0 = (eo < rem) ? relu_bits(d0[i]) : 0.f;
                    const float m1 = (eo + 32 < rem) ? relu_bits(d1[i]) : 0.f;
                    acc[i] = (acc[i] + m0) + m1;
                }
            }
        }
    };

    // Two chunks (128 edges, 3.6 KB of loads per wave) are requested before the first one is consumed: first the two
    // target-id loads, then the edge state, then -- one wait later -- the two Pd gathers.  The loads are unconditional
    // (addresses are clamped to the segment) so that they stay in one basic block and the compiler can wait for
    // them chunk by chunk: they return in order, chunk 0 is computed while chunk 1 is still in flight.
    // (Four chunks per round were measured too: +5 % on 64 x dense256, -5 % on 512 x dense128, 143 VGPRs; not kept.)
    // The FIRST round's target ids and edge state are requested BEFORE the LDS staging stores and the workgroup barrier (they
    // need the CSR offsets only): their round trip runs under the staging wait and the barrier instead of after them.
    const int stride = 64 * wps;
    int base = seg_s + 64 * sub;
    Chunk c0, c1;
    load_index(base, c0);
    load_index(base + stride, c1);
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would cluster these loads in its own order)
    request_node_operands();
    __builtin_amdgcn_sched_barrier(0);
    load_state(base, c0);
    load_state(base + stride, c1);
    // the first round's P_dst gathers go out before the staging stores too (not when they read the LDS table): with computed target ids
    // they need no wait at all, and the staging wait -- a full round trip for the first loads of the launch -- runs under them
    __builtin_amdgcn_sched_barrier(0);   // (and would put the gathers' address arithmetic -- a wait for the ids -- ahead of the state requests)
    if (!PD_LDS) {
        load_target(c0);
        load_target(c1);
    }
    if (MSG) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        l4[tid] = stage_proj[0];
        if (tid + 256 < kH * kProjOut / 4) l4[tid + 256] = stage_proj[1];
    }
    if (PD_LDS) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_pd);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (tid + i * 256 < pd_n4) l4[tid + i * 256] = stage_pd[i];
    }
    GNNCCA_STAMP(p.stamp_slot, 1);
    // INVARIANT (shared with mpn_step_pipe_kernel): between the staging stores above and the combine's __syncthreads() below NOTHING reads
    // s_proj or s_part, and no wave returns or skips that barrier (the BAD_INDEX return above is block-uniform and precedes the stores).
    // With several waves per node the cross-wave combine's barrier is therefore the one that publishes s_proj to the epilogue, its only
    // reader, and no early barrier is needed unless the gathers read s_pd.  A new LDS read in between, or a per-wave early exit, turns
    // this into a silent race: GNNCCA_STEP_EARLYBAR (diag bit 3) restores the early barrier to bisect such a change.
    if (PD_LDS || (MSG && (wps == 1 || (p.diag & 8)))) __syncthreads();   // (diag bit 3: A/B with the early barrier of rounds 1-2)
    GNNCCA_STAMP(p.stamp_slot, 2);
    auto round_body = [&](int rb, Chunk& a, Chunk& b, bool requested = false) {
        if (!requested) {
            load_target(a);
            load_target(b);
        }
        if (DERIVE && p.rng != nullptr) {
            derive(rb, a);
            if (rb + stride < seg_t) derive(rb + stride, b);
        }
        GNNCCA_STAMP(p.stamp_slot, 3);
        compute_chunk(rb, a);
        if (rb + stride < seg_t) compute_chunk(rb + stride, b);
        GNNCCA_STAMP(p.stamp_slot, 4);
    };
    if (base < seg_t) {
        // the SECOND round's target ids (2 registers) are requested before the first round is computed: when that round comes,
        // its edge state and its P_dst gather go out together -- one round trip instead of two (a 255-edge segment owned by one
        // wave has exactly two rounds)
        const int base2 = base + 2 * stride;
        Chunk d0, d1;
        if (!PD_LDS) {
            load_index(base2, d0);
            load_index(base2 + stride, d1);
        }
        round_body(base, c0, c1, !PD_LDS);
        if (base2 < seg_t) {
            if (PD_LDS) {
                load_index(base2, d0);
                load_index(base2 + stride, d1);
            }
            load_state(base2, d0);
            load_state(base2 + stride, d1);
            round_body(base2, d0, d1);
        }
        base += 4 * stride;
    }
    for (; base < seg_t; base += 2 * stride) {
        load_index(base, c0);
        load_index(base + stride, c1);
        load_state(base, c0);
        load_state(base + stride, c1);
        round_body(base, c0, c1);
    }
    GNNCCA_STAMP(p.stamp_slot, 5);

    if (MSG) {
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
        if (DERIVE && p.rng != nullptr && active && sub == 0 && lane == 0) {   // (start1, len1, start2 - len1, breaks) of this node
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const int len1 = d_nb ? d_first : seg_t - seg_s;
            reinterpret_cast<i32x4*>(p.rng)[node] = i32x4{d_start1, len1, d_start2 - len1, d_nb};
            if (d_nb > 1) atomicOr(p.flags + 1, 1u);   // not two runs: every later step of this forward streams col32
        }
        GNNCCA_STAMP(p.stamp_slot, 6);
        if (active && sub == 0) {
            const int deg = seg_t - seg_s;
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
                p.pd_out[(size_t)node * kPdStride + lane] = pr;
            else if (lane < kProjOut)
                p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
        }
    }
    GNNCCA_STAMP(p.stamp_slot, 7);
}

template <bool FIRST, bool CLS, bool MSG, bool PDL, bool EB, int NT>
static hipError_t launch_fast_t(const StepParams& sp, hipStream_t st) {
    const int npg = 4 / sp.wps;
    const unsigned blocks = (unsigned)((sp.N + npg - 1) / npg);
    const size_t lds = ((MSG ? (size_t)kH * kProjOut : 0) + 4 * kH + 16 + (PDL ? (size_t)sp.N * kPdStride : 0)) * sizeof(float);
    GNNCCA_LAUNCH((mpn_step_fast_kernel<FIRST, CLS, MSG, PDL, EB, NT>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool FIRST, bool CLS, bool MSG, bool PDL>
static hipError_t launch_fast(const StepParams& sp, hipStream_t st) {
    if (!PDL) {   // the non-temporal variants only exist beyond the LDS-resident gather table (N > 1024): big batches
        const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
        if (nt == 2) return sp.e_bf16 ? launch_fast_t<FIRST, CLS, MSG, false, true, 2>(sp, st) : launch_fast_t<FIRST, CLS, MSG, false, false, 2>(sp, st);
        if (nt == 1) return sp.e_bf16 ? launch_fast_t<FIRST, CLS, MSG, false, true, 1>(sp, st) : launch_fast_t<FIRST, CLS, MSG, false, false, 1>(sp, st);
    }
    return sp.e_bf16 ? launch_fast_t<FIRST, CLS, MSG, PDL, true, 0>(sp, st) : launch_fast_t<FIRST, CLS, MSG, PDL, false, 0>(sp, st);
}

#ifndef GNNCCA_KERNELS_ONLY   // (tools: compile-only probes of single instantiations skip the dispatch tables)
static hipError_t launch_fast_dispatch(const StepParams& sp, bool msg, hipStream_t st) {
    const int key = (sp.first ? 8 : 0) | (sp.cls_layers ? 4 : 0) | (msg ? 2 : 0) | (sp.pd_lds ? 1 : 0);
    switch (key) {
#define GNNCCA_FAST_CASE(K, A, B, C, D) \
    case K: return launch_fast<A, B, C, D>(sp, st);
        GNNCCA_FAST_CASE(0, false, false, false, false)
        GNNCCA_FAST_CASE(1, false, false, false, true)
        GNNCCA_FAST_CASE(2, false, false, true, false)
        GNNCCA_FAST_CASE(3, false, false, true, true)
        GNNCCA_FAST_CASE(4, false, true, false, false)
        GNNCCA_FAST_CASE(5, false, true, false, true)
        GNNCCA_FAST_CASE(6, false, true, true, false)
        GNNCCA_FAST_CASE(7, false, true, true, true)
        GNNCCA_FAST_CASE(8, true, false, false, false)
        GNNCCA_FAST_CASE(9, true, false, false, true)
        GNNCCA_FAST_CASE(10, true, false, true, false)
        GNNCCA_FAST_CASE(11, true, false, true, true)
        GNNCCA_FAST_CASE(12, true, true, false, false)
        GNNCCA_FAST_CASE(13, true, true, false, true)
        GNNCCA_FAST_CASE(14, true, true, true, false)
        GNNCCA_FAST_CASE(15, true, true, true, true)
#undef GNNCCA_FAST_CASE
    }
    return hipErrorInvalidValue;
}
#endif


}  // namespace gnncca
