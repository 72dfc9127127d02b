#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// PERSISTENT form of mpn_step_fast_kernel for big batches (N >= 8192 nodes, one wave per source-node segment).
// Same per-chunk arithmetic, same layouts, same results bit for bit; what changes is a wave's life.  In the
// one-node-per-wave form every wave pays, per node: a workgroup launch, the 6 KB projection matrix into LDS and a
// barrier, then two dependent round trips (CSR offsets + the node's P_src / Q row, then edge state + target ids)
// before its first FMA -- and the four waves of a SIMD, launched together, walk through those phases in lock-step:
// they queue for the VALU and the matrix pipe at the same moments and leave both idle while they all wait on memory
// (counters, 64 x dense256: waves 43 % issue-stalled, 34 % parked on s_waitcnt, 23 % issuing; MFMA pipe 19 %, VALU
// 44 % busy; halving the bytes with a bf16 edge state bought 8 %).  Here a workgroup stays resident (grid = 4 per
// CU), stages the projection matrix ONCE, and each wave walks nodes gw, gw + G, gw + 2G, ... with the NEXT node's
// CSR offsets and P_src / Q row already in flight while it works on the current one; the waves drift apart after
// the first node, so one computes while its SIMD partners wait.
// ------------------------------------------------------------------------------------------------------------
template <bool FIRST, bool CLS, bool MSG, bool EBF16>
#ifndef GNNCCA_PERSIST_WPC
#define GNNCCA_PERSIST_WPC 3   // resident workgroups per CU = waves per SIMD the register budget allows
#endif
__global__ __launch_bounds__(256, GNNCCA_PERSIST_WPC) void mpn_step_persist_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_pd = smem;                                     // unused here (the gather table stays in L2 at these sizes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);
    const int half = lane >> 5, ch = lane & 31;

    // ---- once per workgroup ---------------------------------------------------------------------------------------------------
    const unsigned gflags = p.flags[0];
    const int G = gridDim.x * 4;
    int node = blockIdx.x * 4 + wave;
    // prologue of a node: CSR offsets and its (P_src | Q) row -- requested one node ahead
    int nx_s = 0, nx_t = 0;
    float nx_psrc[kEF], nx_cinit = 0.f;
    auto request_node = [&](int nd) {
        const int nc = min(nd, p.N - 1);
        nx_s = p.seg_ptr[nc];
        nx_t = p.seg_ptr[nc + 1];
        const float* __restrict__ psq = p.psq_in + (size_t)nc * kPsQStride;
#pragma unroll
        for (int f = 0; f < kEF; ++f) nx_psrc[f] = psq[f];
        if (MSG) nx_cinit = psq[8 + ch];
    };
    request_node(node);
    float bw[3] = {0.f, 0.f, 0.f};
    float projb_l = 0.f;
    if (MSG) {
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
#pragma unroll
        for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        const f32x4 s0 = g4[tid], s1 = g4[min(tid + 256, kH * kProjOut / 4 - 1)];   // 384 float4 in all
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        l4[tid] = s0;
        if (tid + 256 < kH * kProjOut / 4) l4[tid + 256] = s1;
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    const bool unsorted = (gflags & GNNCCA_GRAPH_UNSORTED) != 0;
    const bool padded = p.ell_S > 0 && !(gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR));
    if (MSG) __syncthreads();   // the only barrier: the waves are independent from here on

    for (; node < p.N; node += G) {
        const int seg_s = nx_s, seg_t = nx_t;
        float psrc[kEF];
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = nx_psrc[f];
        const float cinit = nx_cinit;
        request_node(node + G);   // in flight while this node is worked on (addresses clamped; the values of a node >= N are unused)
        const long long eoff = padded ? (long long)node * p.ell_S - seg_s : 0ll;   // see mpn_step_fast_kernel
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int last = seg_t - 1;

        struct Chunk {
            float raw[kEF];
            float pd[kEF];
            int ko, j;
        };
        // phase A: everything addressed by the edge slot itself (target id, permutation, edge state)
        auto load_index = [&](int base, Chunk& c) {
            const int kk = min(base + lane, last);
            c.ko = unsorted ? p.perm[kk] : kk;
            c.j = p.col32[kk];
        };
        auto load_state = [&](int base, Chunk& c) {
            const int kk = min(base + lane, last);
            if (FIRST) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.edge_attr + (size_t)c.ko * 4);
                c.raw[0] = a[0], c.raw[1] = a[1], c.raw[2] = a[2], c.raw[3] = a[3], c.raw[4] = 0.f, c.raw[5] = 0.f;
            } else if (EBF16) {
                // edge state stored as three planes of packed bf16 pairs: one dword load = two features
                const unsigned* __restrict__ e2 = reinterpret_cast<const unsigned*>(p.e);
    #pragma unroll
                for (int f = 0; f < kEF / 2; ++f) {
                    const unsigned w = e2[(size_t)f * p.e_stride + kk + eoff];
                    c.raw[2 * f] = __uint_as_float(w << 16);
                    c.raw[2 * f + 1] = __uint_as_float(w & 0xFFFF0000u);
                }
            } else {
    #pragma unroll
                for (int f = 0; f < kEF; ++f) c.raw[f] = p.e[(size_t)f * p.e_stride + kk + eoff];
            }
        };
        // phase B: the gather that depends on the target id
        auto load_target = [&](Chunk& c) {
            f32x4 a;
            f32x2 b2;
            if (false) {
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
                        e2[(size_t)f * p.e_stride + k + eoff] = __builtin_bit_cast(unsigned, pk);
                    }
                } else {
    #pragma unroll
                    for (int f = 0; f < kEF; ++f) p.e[(size_t)f * p.e_stride + k + eoff] = en[f];
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
                if (valid) p.logits[c.ko] = logit;
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
                    for (int i = 0; i < 16; ++i) acc[i] += relu_bits(d0[i]) + relu_bits(d1[i]);
                } else {
                    const int rem = seg_t - base - 4 * half;
    #pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int eo = (i & 3) + 8 * (i >> 2);
                        const float m0 = (eo < rem) ? relu_bits(d0[i]) : 0.f;
                        const float m1 = (eo + 32 < rem) ? relu_bits(d1[i]) : 0.f;
                        acc[i] += m0 + m1;
                    }
                }
            }
        };

        // Two chunks (128 edges, 3.6 KB of loads per wave) are requested before the first one is consumed: first the two
        // target-id loads, then the edge state, then -- one wait later -- the two Pd gathers.  The loads are unconditional
        // (addresses are clamped to the segment) so that they stay in one basic block and the compiler can wait for
        // them chunk by chunk: they return in order, chunk 0 is computed while chunk 1 is still in flight.
        // (Four chunks per round were measured too: +5 % on 64 x dense256, -5 % on 512 x dense128, 143 VGPRs; not kept.)
        const int stride = 64;
        for (int base = seg_s; base < seg_t; base += 2 * stride) {
            Chunk c0, c1;
            const bool two = base + stride < seg_t;
            load_index(base, c0);
            load_index(base + stride, c1);
            load_state(base, c0);
            load_state(base + stride, c1);
            load_target(c0);
            load_target(c1);
            compute_chunk(base, c0);
            if (two) compute_chunk(base + stride, c1);
        }

        if (MSG) {
            float v = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += acc[i];
            v += __shfl_xor(v, 32);
            const int deg = seg_t - seg_s;
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
        }
    }
}

template <bool FIRST, bool CLS, bool MSG>
static hipError_t launch_persist(const StepParams& sp, hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<long long>(((long long)sp.N + 3) / 4, 256 * GNNCCA_PERSIST_WPC);   // all resident
    const size_t lds = (MSG ? (size_t)kH * kProjOut : 0) * sizeof(float) + 64;
    if (sp.e_bf16)
        GNNCCA_LAUNCH((mpn_step_persist_kernel<FIRST, CLS, MSG, true>), dim3(blocks), dim3(256), lds, st, sp);
    else
        GNNCCA_LAUNCH((mpn_step_persist_kernel<FIRST, CLS, MSG, false>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

static hipError_t launch_persist_dispatch(const StepParams& sp, bool msg, hipStream_t st) {
    const int key = (sp.first ? 4 : 0) | (sp.cls_layers ? 2 : 0) | (msg ? 1 : 0);
    switch (key) {
        case 0: return launch_persist<false, false, false>(sp, st);
        case 1: return launch_persist<false, false, true>(sp, st);
        case 2: return launch_persist<false, true, false>(sp, st);
        case 3: return launch_persist<false, true, true>(sp, st);
        case 4: return launch_persist<true, false, false>(sp, st);
        case 5: return launch_persist<true, false, true>(sp, st);
        case 6: return launch_persist<true, true, false>(sp, st);
        case 7: return launch_persist<true, true, true>(sp, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace gnncca
