#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Step kernel for BIG, (nearly) REGULAR batches -- 64 x dense256, 512 x dense128, dense1024: the operating points where
// the step is bound by HBM/fabric bytes, not by latency.  Same algebra and the same fused work per edge as
// mpn_step_fast_kernel; what changes is the layout of the edge state and the access width:
//
//   * PADDED layout ("ELL"): node i owns the slots [i*S, (i+1)*S) of every feature plane, S = 128 * units, chosen by the
//     host from ceil(E/N) alone (no device read-back).  Every segment starts on a 512-B boundary, so no 128-B line of
//     the edge state is shared by two segments (with the compact CSR layout 19-27 % more bytes crossed the fabric than
//     the algorithm needs: lines straddling two segments are fetched and written back twice).
//   * a lane owns FOUR consecutive slots: every plane access is one 16-B load / store per lane (dwordx4, the width the
//     chip streams fastest at), 4x fewer memory instructions and address computations per edge.
//   * a HALF-wave (32 lanes x 4 slots = 128 slots) is the unit of work; a workgroup owns 8 units = 8, 4, 2 or 1 whole
//     node(s) (or loops when a node has more than 1024 slots), so the per-destination reduction still needs no atomics.
//     Half h of a wave feeds MFMA tile d_h: the sum over a tile's rows is the sum over one unit's edges.
//   * padding slots hold e = 0 and the target id N, a SENTINEL row of the P_dst table filled with -3e38: the edge update
//     of a padding slot is ReLU(-3e38 + ...) = 0 again, with no select in the hot loop; the bias Q[node] enters the MFMA
//     through a fourth k-step whose A operand is the slot's validity (1 / 0), so a padding row of the tile is exactly 0.
//
// The host cannot know the maximum degree (it never reads the graph back): the plan kernel checks deg <= S per node and
// raises GNNCCA_GRAPH_IRREGULAR otherwise.  This kernel then returns at once and the CSR kernel, which is launched
// behind it in this regime and returns at once when the flag is clear, does the step (also for unsorted rows).
// ------------------------------------------------------------------------------------------------------------
constexpr float kPdSentinel = -3.0e38f;

struct __attribute__((packed, aligned(4))) f32x4_u4 {  // four floats at a 4-byte aligned address
    float v[4];
};

__device__ __forceinline__ float relu_bits(float x) {
    // max as a signed integer: negative floats (sign bit set) are negative integers.  One VALU op; fmaxf would first
    // canonicalise the MFMA result (a second v_max per element)
    return __int_as_float(max(__float_as_int(x), 0));
}

#ifdef GNNCCA_WIDE_WAVES   // diagnostic builds: force the register budget of N waves per SIMD
#define GNNCCA_WIDE_ATTR __attribute__((amdgpu_waves_per_eu(GNNCCA_WIDE_WAVES, GNNCCA_WIDE_WAVES)))
#else
#define GNNCCA_WIDE_ATTR
#endif
template <bool FIRST, bool CLS, bool MSG, bool PD_LDS, bool EBF16, bool TWO>
__global__ __launch_bounds__(256) GNNCCA_WIDE_ATTR void mpn_step_wide_kernel(const StepParams p) {
    // TWO: the two halves of a wave belong to two different nodes (one 128-slot unit per node, ell_U == 1)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                   // [32][48]   (MSG)
    float* s_part = s_proj + (MSG ? kH * kProjOut : 0);     // [4][32]
    float* s_pd = s_part + 4 * kH;                          // [N + 1][8] (PD_LDS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);

    const int U = TWO ? 1 : p.ell_U, S = p.ell_S;           // units (of 128 slots) a node gets per pass; slots per node
    const int half = lane >> 5, ch = lane & 31;
    const int hw = 2 * wave + half;                         // half-wave of the workgroup, 0..7
    const int node = blockIdx.x * (8 / U) + hw / U;         // per HALF (the two halves differ when TWO)
    const int usub = hw % U;
    const bool active = node < p.N;
    const int nclamp = active ? node : 0;
    const size_t slot0 = (size_t)(active ? node : p.N) * S + usub * 128 + 4 * ch;  // halves without a node: the dump area

    // ---- the edge state of the first pass: its addresses depend on nothing but the thread's position, so these loads go
    // out before everything else (with the compact layout they had to wait for the CSR offsets) -------------------------
    float raw[4][kEF];
    int j[4];
    auto load_state = [&](size_t slot) {
        if (EBF16) {
            const unsigned* __restrict__ e2 = reinterpret_cast<const unsigned*>(p.e);
#pragma unroll
            for (int f = 0; f < kEF / 2; ++f) {
                const uint4 w = *reinterpret_cast<const uint4*>(e2 + (size_t)f * p.e_stride + slot);
                const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    raw[q][2 * f] = __uint_as_float(ww[q] << 16);
                    raw[q][2 * f + 1] = __uint_as_float(ww[q] & 0xFFFF0000u);
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.e + (size_t)f * p.e_stride + slot);
                raw[0][f] = v[0], raw[1][f] = v[1], raw[2][f] = v[2], raw[3][f] = v[3];
            }
        }
        const int4 jj = *reinterpret_cast<const int4*>(p.col_ell + slot);
        j[0] = jj.x, j[1] = jj.y, j[2] = jj.z, j[3] = jj.w;
    };
    if (!FIRST) load_state(slot0);

    // ---- prologue: every other independent load is issued before the first wait ---------------------------------------
    const unsigned gflags = p.flags[0];
    const int seg_s = p.seg_ptr[nclamp];
    const int seg_t = p.seg_ptr[nclamp + 1];
    const float* __restrict__ psq = p.psq_in + (size_t)nclamp * kPsQStride;
    float psrc[kEF];
#pragma unroll
    for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
    float q_own = 0.f;
    float bw[3] = {0.f, 0.f, 0.f};
    f32x4 stage_proj[2];
    f32x4 stage_pd[9];
    float projb_l = 0.f;
    if (MSG) {
        q_own = psq[8 + ch];
        projb_l = blob[p.off_projb + min(lane, kProjOut - 1)];
#pragma unroll
        for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
        stage_proj[0] = g4[tid];
        stage_proj[1] = g4[min(tid + 256, kH * kProjOut / 4 - 1)];
    }
    const int pd_n4 = (p.N + 1) * (kPdStride / 4);          // with the sentinel row
    if (PD_LDS) {  // N <= 1024: at most 9 float4 per thread
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(p.pd_in);
#pragma unroll
        for (int i = 0; i < 9; ++i) stage_pd[i] = g4[min(tid + i * 256, pd_n4 - 1)];
    }
    if (MSG) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_proj);
        l4[tid] = stage_proj[0];
        if (tid + 256 < kH * kProjOut / 4) l4[tid + 256] = stage_proj[1];
    }
    if (PD_LDS) {
        f32x4* l4 = reinterpret_cast<f32x4*>(s_pd);
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (tid + i * 256 < pd_n4) l4[tid + i * 256] = stage_pd[i];
    }
    if (gflags & GNNCCA_GRAPH_BAD_INDEX) {
        if (CLS)
            for (size_t k = (size_t)blockIdx.x * 256 + tid; k < (size_t)p.E; k += (size_t)gridDim.x * 256)
                p.logits[k] = __builtin_nanf("");
        return;
    }
    if (gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR)) return;  // the CSR kernel behind this launch does the step
    if (MSG || PD_LDS) __syncthreads();

    const int deg = active ? seg_t - seg_s : 0;
    const int last_k = max(p.E - 1, 0);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    // MFMA B operand of the bias k-step: B[k][n] sits in lane n + 32 k; k = 0 carries Q[node of the tile's half][n], k = 1 is
    // multiplied by A[.][1] = 0 and must only be finite
    float bq0 = 0.f, bq1 = 0.f;
    if (MSG) {
        const float q_other = __shfl_xor(q_own, 32);
        bq0 = half == 0 ? q_own : 0.f;
        bq1 = half == 0 ? q_other : 0.f;
    }
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

    const int iters = S / (128 * U);
    for (int it = 0; it < iters; ++it) {
        const int lb = (it * U + usub) * 128 + 4 * ch;      // my first slot inside the segment
        if (it > 0 && !__any((it * U + usub) * 128 < deg)) continue;  // both halves past their segments: slots stay as they are
        const size_t slot = slot0 + (size_t)it * U * 128;
        // ---- loads: the four slots of the lane, all planes ---------------------------------------------------------------
        if (FIRST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = min(seg_s + lb + q, last_k);
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.edge_attr + (size_t)kk * 4);
                raw[q][0] = a[0], raw[q][1] = a[1], raw[q][2] = a[2], raw[q][3] = a[3], raw[q][4] = 0.f, raw[q][5] = 0.f;
                j[q] = p.col32[kk];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = (lb + q < deg) ? j[q] : p.N;   // padding slots point at the sentinel row
        } else if (it > 0) {
            load_state(slot);
        }
        float pdn[kEF];   // P_dst row of the NEXT slot: gathered one slot ahead of its use (12 registers instead of 24)
        auto gather = [&](int jq, float (&dst)[kEF]) {
            f32x4 a;
            f32x2 b2;
            if (PD_LDS) {
                a = *reinterpret_cast<const f32x4*>(s_pd + jq * kPdStride);
                b2 = *reinterpret_cast<const f32x2*>(s_pd + jq * kPdStride + 4);
            } else {
                const float* __restrict__ pdj = p.pd_in + (size_t)jq * kPdStride;
                a = *reinterpret_cast<const f32x4*>(pdj);
                b2 = *reinterpret_cast<const f32x2*>(pdj + 4);
            }
            dst[0] = a[0], dst[1] = a[1], dst[2] = a[2], dst[3] = a[3], dst[4] = b2[0], dst[5] = b2[1];
        };
        gather(j[0], pdn);
        if (FIRST && p.store_e) {
            int4 jj;
            jj.x = j[0], jj.y = j[1], jj.z = j[2], jj.w = j[3];
            *reinterpret_cast<int4*>(p.col_ell + slot) = jj;
        }
        float en[4][kEF];
        float lg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float pdq[kEF];
#pragma unroll
            for (int f = 0; f < kEF; ++f) pdq[f] = pdn[f];
            if (q < 3) gather(j[q + 1], pdn);
            float ein[kEF];
            if (FIRST) {
#pragma unroll
                for (int h = 0; h < kEF / 2; ++h) {
                    f32x2 s = {cw[kFcEncB + 2 * h], cw[kFcEncB + 2 * h + 1]};
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x2 w = {cw[kFcEncW + a * kEF + 2 * h], cw[kFcEncW + a * kEF + 2 * h + 1]};
                        s = __builtin_elementwise_fma(w, f32x2{raw[q][a], raw[q][a]}, s);
                    }
                    ein[2 * h] = fmaxf(s[0], 0.f), ein[2 * h + 1] = fmaxf(s[1], 0.f);
                }
            } else {
#pragma unroll
                for (int f = 0; f < kEF; ++f) ein[f] = raw[q][f];
            }
            f32x2 s2[kEF / 2];
#pragma unroll
            for (int h = 0; h < kEF / 2; ++h) s2[h] = f32x2{psrc[2 * h], psrc[2 * h + 1]} + f32x2{pdq[2 * h], pdq[2 * h + 1]};
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
            for (int h = 0; h < kEF / 2; ++h) en[q][2 * h] = fmaxf(s2[h][0], 0.f), en[q][2 * h + 1] = fmaxf(s2[h][1], 0.f);
            if (CLS) {
                f32x2 z[2] = {f32x2{cw[kFcCb1], cw[kFcCb1 + 1]}, f32x2{cw[kFcCb1 + 2], cw[kFcCb1 + 3]}};
#pragma unroll
                for (int f = 0; f < kEF; ++f) {
                    const f32x2 x = {en[q][f], en[q][f]};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 w = {cw[kFcCw1 + f * 4 + 2 * h], cw[kFcCw1 + f * 4 + 2 * h + 1]};
                        z[h] = __builtin_elementwise_fma(w, x, z[h]);
                    }
                }
                float logit = cw[kFcCb2];
#pragma unroll
                for (int u = 0; u < 4; ++u) logit = fmaf(cw[kFcCw2 + u], fmaxf(z[u >> 1][u & 1], 0.f), logit);
                lg[q] = logit;
            }
            if (MSG) {
                // rows of tile h = the 32 lanes of half h (slot q of each): A[m][k] sits in lane m + 32 k.  The bias Q enters
                // through a k-step of its own whose A operand is the slot's validity, so a padding row is exactly 0.
                const float vf = (lb + q < deg) ? 1.f : 0.f;
                const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(vf), 0u, false, false);
                unsigned a_op[3][2];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(en[q][2 * s]), __float_as_uint(en[q][2 * s + 1]),
                                                                    false, false);
                    a_op[s][0] = r[0], a_op[s][1] = r[1];
                }
                // one tile after the other through the SAME sixteen registers
                f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(rb[0]), bq0, zero16, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 3; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a_op[s][0]), bw[s], d, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc0[i] += relu_bits(d[i]);
                asm volatile("" : "+v"(acc0));
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(rb[1]), bq1, zero16, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 3; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a_op[s][1]), bw[s], d, 0, 0, 0);
                if (TWO) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc1[i] += relu_bits(d[i]);
                    asm volatile("" : "+v"(acc1));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc0[i] += relu_bits(d[i]);
                    asm volatile("" : "+v"(acc0));
                }
            }
            // one slot after the other: without the pins above and this barrier the compiler interleaves the four slots' MFMA
            // chains, sinks the accumulations below the loop and keeps eight accumulator tiles alive (248 VGPRs)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- stores: whole 16-B pieces, padding slots included (they hold exact zeros) -------------------------------
        if (p.store_e) {
            if (EBF16) {
                unsigned* __restrict__ e2 = reinterpret_cast<unsigned*>(p.e);
#pragma unroll
                for (int f = 0; f < kEF / 2; ++f) {
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    unsigned ww[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x2_t pk;  // round to nearest even (v_cvt_pk_bf16_f32)
                        pk[0] = (__bf16)en[q][2 * f];
                        pk[1] = (__bf16)en[q][2 * f + 1];
                        ww[q] = __builtin_bit_cast(unsigned, pk);
                    }
                    uint4 w;
                    w.x = ww[0], w.y = ww[1], w.z = ww[2], w.w = ww[3];
                    *reinterpret_cast<uint4*>(e2 + (size_t)f * p.e_stride + slot) = w;
                }
            } else {
#pragma unroll
                for (int f = 0; f < kEF; ++f) {
                    const f32x4 v = {en[0][f], en[1][f], en[2][f], en[3][f]};
                    *reinterpret_cast<f32x4*>(p.e + (size_t)f * p.e_stride + slot) = v;
                }
            }
        }
        if (CLS) {
            float* __restrict__ dst = p.logits + seg_s + lb;   // the caller's (compact) edge order: 4-byte aligned only
            if (lb + 3 < deg) {
                f32x4_u4 v;
                v.v[0] = lg[0], v.v[1] = lg[1], v.v[2] = lg[2], v.v[3] = lg[3];
                *reinterpret_cast<f32x4_u4*>(dst) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lb + q < deg) dst[q] = lg[q];
            }
        }
    }

    if (MSG) {
        // registers -> both halves of the wave (lanes c and c + 32 hold channel c of every tile)
        float v0 = acc0[0], v1 = acc1[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) v0 += acc0[i], v1 += acc1[i];
        v0 += __shfl_xor(v0, 32);   // the node of half 0 (of the whole wave unless TWO)
        v1 += __shfl_xor(v1, 32);   // TWO: the node of half 1
        const float* w = s_proj + min(lane, kProjOut - 1);
        auto project = [&](float v, int nd, int dg) {   // nd, dg wave-uniform
            if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(dg, 1);
            if (dg == 0) v = 0.f;
            float pr = projb_l;
#pragma unroll
            for (int c = 0; c < kH; ++c)
                pr = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)), pr);
            if (lane < kPdStride)
                p.pd_out[(size_t)nd * kPdStride + lane] = pr;
            else if (lane < kProjOut)
                p.psq_out[(size_t)nd * kPsQStride + lane - kPdStride] = pr;
        };
        const int nd0 = __builtin_amdgcn_readlane(node, 0), nd1 = __builtin_amdgcn_readlane(node, 32);
        const int dg0 = __builtin_amdgcn_readlane(deg, 0), dg1 = __builtin_amdgcn_readlane(deg, 32);
        if (TWO) {
            if (nd0 < p.N) project(v0, nd0, dg0);
            if (nd1 < p.N) project(v1, nd1, dg1);
        } else {
            float v = v0;
            const int wpn = U / 2;                      // waves per node: 1, 2 or 4
            if (wpn > 1) {
                if (lane < kH) s_part[wave * kH + lane] = v;
                __syncthreads();
                if (wave % wpn == 0) {
                    v = s_part[wave * kH + ch];
                    for (int u = 1; u < wpn; ++u) v += s_part[(wave + u) * kH + ch];
                }
            }
            if (wave % wpn == 0 && nd0 < p.N) project(v, nd0, dg0);
        }
        if (blockIdx.x == 0 && tid < kPdStride) p.pd_out[(size_t)p.N * kPdStride + tid] = kPdSentinel;
    }
}

template <bool FIRST, bool CLS, bool MSG, bool PDL, bool EB>
static hipError_t launch_wide_t(const StepParams& sp, hipStream_t st) {
    const int npw = 8 / sp.ell_U;
    const unsigned blocks = (unsigned)((sp.N + npw - 1) / npw);
    const size_t lds = ((MSG ? (size_t)kH * kProjOut : 0) + 4 * kH + (PDL ? (size_t)(sp.N + 1) * kPdStride : 0)) * sizeof(float);
    if (sp.ell_U == 1)
        GNNCCA_LAUNCH((mpn_step_wide_kernel<FIRST, CLS, MSG, PDL, EB, true>), dim3(blocks), dim3(256), lds, st, sp);
    else
        GNNCCA_LAUNCH((mpn_step_wide_kernel<FIRST, CLS, MSG, PDL, EB, false>), dim3(blocks), dim3(256), lds, st, sp);
    return hipGetLastError();
}

template <bool FIRST, bool CLS, bool MSG, bool PDL>
static hipError_t launch_wide(const StepParams& sp, hipStream_t st) {
    return sp.e_bf16 ? launch_wide_t<FIRST, CLS, MSG, PDL, true>(sp, st) : launch_wide_t<FIRST, CLS, MSG, PDL, false>(sp, st);
}

static hipError_t launch_wide_dispatch(const StepParams& sp, bool msg, hipStream_t st) {
    const int key = (sp.first ? 8 : 0) | (sp.cls_layers ? 4 : 0) | (msg ? 2 : 0) | (sp.pd_lds ? 1 : 0);
    switch (key) {
#define GNNCCA_WIDE_CASE(K, A, B, C, D) \
    case K: return launch_wide<A, B, C, D>(sp, st);
        GNNCCA_WIDE_CASE(0, false, false, false, false)
        GNNCCA_WIDE_CASE(1, false, false, false, true)
        GNNCCA_WIDE_CASE(2, false, false, true, false)
        GNNCCA_WIDE_CASE(3, false, false, true, true)
        GNNCCA_WIDE_CASE(4, false, true, false, false)
        GNNCCA_WIDE_CASE(5, false, true, false, true)
        GNNCCA_WIDE_CASE(6, false, true, true, false)
        GNNCCA_WIDE_CASE(7, false, true, true, true)
        GNNCCA_WIDE_CASE(8, true, false, false, false)
        GNNCCA_WIDE_CASE(9, true, false, false, true)
        GNNCCA_WIDE_CASE(10, true, false, true, false)
        GNNCCA_WIDE_CASE(11, true, false, true, true)
        GNNCCA_WIDE_CASE(12, true, true, false, false)
        GNNCCA_WIDE_CASE(13, true, true, false, true)
        GNNCCA_WIDE_CASE(14, true, true, true, false)
        GNNCCA_WIDE_CASE(15, true, true, true, true)
#undef GNNCCA_WIDE_CASE
    }
    return hipErrorInvalidValue;
}

}  // namespace gnncca
