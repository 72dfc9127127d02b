#pragma once
// SHELVED EXPERIMENT (round 2) -- not compiled into the library.  Measured on one MI355X (profiles/r02_logs/r2_exp_ring*.log,
// r2_stamps_ring1.log, r2_ubench_ring1.log), 64 x dense256, step launch with message block:
//     mpn_step_fast_kernel 50-51 us;  this kernel 67 us (first form), 75-96 us (gather issued after the use of the previous one),
//     79-81 us (80 VGPRs, four 5-wave workgroups per CU);  last step (no message block) 27.5-29 vs 23-24 us;  512 x dense128 204-214 vs
//     111-127 us.  Logits equal to the shipped kernel's to 6e-8.
// Why it loses: (i) tools/ubench_ring.hip -- the same bytes with trivial arithmetic stream in 33 us (6.5 TB/s) through this ring AND
// through plain one-workgroup-per-item register loads: the memory side of the shipped kernel is not what holds it at 50 us, so a
// better loader cannot win more than the latency it hides; (ii) a 5-wave workgroup fits 2-3 times per CU (the worst SIMD takes two
// of its waves), not 4; (iii) one s_barrier per 256 edges locks four consumer waves and the loader into rounds of ~4300 cycles of
// dependent latency (LDS header -> LDS ids -> gather issue -> edge update -> stores -> classifier -> six dependent MFMAs -> ReLU /
// accumulate), of which ~1000 are VALU issue; (iv) hipcc waits vmcnt(0) for a loop-carried gather, i.e. for the round's stores too.
// It was wired in as: launch_ring(sp, msg, st) for steps 2..L when ell_S >= 96 and N >= 2048, followed by mpn_step_fast_kernel with
// StepParams::ring_skip = 1 (returns at once unless the plan raised a flag; 4 us per step).
// Part of the single translation unit mpn_forward.hip when it was compiled.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Ring form of the specialised step kernel, for big regular batches (padded edge-state layout, S >= 96 slots per node,
// steps 2..L).  The arithmetic per edge is mpn_step_fast_kernel's; what changes is who waits for memory:
//   * workgroups are PERSISTENT (a few per CU) and own a contiguous range of items; an item is 256 consecutive slots of the
//     padded edge state: one node (128 < S <= 256), a 256-slot block of a node (S > 256) or two nodes (S <= 128);
//   * wave 4 is the LOADER: it moves an item's six feature planes (1 KB each), its 256 target ids and its nodes' (P_src | Q)
//     rows into an R-deep ring of LDS slots with LDS-DMA (global_load_lds: no VGPRs, no wait in the issuing wave), R - 1
//     items ahead of the consumers, and publishes them with a counted s_waitcnt vmcnt + the round's one s_barrier;
//   * waves 0-3 are CONSUMERS: per round one 64-edge chunk each from LDS, the P_dst gather for the NEXT item already in flight
//     (its ids are in the ring a round early), then edge update, classifier, MFMA message, stores -- no wait on HBM in steady
//     state;
//   * the cross-wave combine and the per-node projection epilogue move to the loader wave (its projection column lives in 32
//     of its registers), double-buffered partial sums, no extra barrier.
// Order of the per-node sum: four per-wave partials (each wave: its chunks in order, the 16 accumulator registers, the two
// half-waves) added in wave order -- fixed, so results are bitwise reproducible run to run; it differs from the one-wave-per-
// node order of mpn_step_fast_kernel by rounding only.
// Flags (unsorted rows, a degree above S, a bad index) are only known on the device: the kernel then returns at once and the
// launch code's second launch (mpn_step_fast_kernel with ring_skip = 1, which returns at once in the regular case) does the step.
// ------------------------------------------------------------------------------------------------------------
constexpr int kRingCol = 6 * 256;           // [256] target ids, compact order from the item's first edge
constexpr int kRingPsq = kRingCol + 256;    // [2][64] (P_src | Q) rows of the item's nodes (40 floats of each 64 used)
constexpr int kRingHdr = kRingPsq + 128;    // ints: seg_ptr[node0], [node0 + 1], [node0 + 2]
constexpr int kRingSlot = kRingHdr + 16;    // floats per slot (7744 B)
constexpr int kRingDma = 12;                // LDS-DMA instructions per item: 6 planes + 4 x 64 ids + 2 rows

__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// one LDS-DMA wave instruction: lane l's 16 (4) bytes at gsrc land at lds_dst + 16 (4) * l; M0 carries the LDS base and is
// written in the statement that reads it (the compiler owns M0 otherwise)
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Diagnostic build (-DGNNCCA_STAMPS, tools/stamps_ring.py): per-wave totals of the cycles spent in each phase of a round.
#ifdef GNNCCA_STAMPS
#define RING_T_DECL unsigned long long rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt_last = __builtin_amdgcn_s_memtime()
#define RING_T(i)                                                      \
    do {                                                               \
        const unsigned long long rt_now = __builtin_amdgcn_s_memtime(); \
        rt_acc[i] += rt_now - rt_last;                                 \
        rt_last = rt_now;                                              \
    } while (0)
#define RING_T_FLUSH(blk, wv)                                                                                 \
    do {                                                                                                      \
        if (g_stamps && lane == 0)                                                                            \
            for (int q = 0; q < 8; ++q) g_stamps[((((size_t)p.stamp_slot) * 4096 + (blk)) * 4 + (wv)) * 16 + q] = rt_acc[q]; \
    } while (0)
#else
#define RING_T_DECL do { } while (0)
#define RING_T(i) do { } while (0)
#define RING_T_FLUSH(blk, wv) do { } while (0)
#endif
#define GNNCCA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define GNNCCA_RING_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool CLS, bool MSG, int NT, int R>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(6, 6))) void mpn_step_ring_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;                       // [R][kRingSlot]
    float* s_part = ring + R * kRingSlot;     // [2][4][32] per-wave partial sums of the node(s) of an item (MSG)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ blob = p.blob;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* cw = (cfloat*)(unsigned long long)(blob + p.off_fast);

    const unsigned gflags = p.flags[0];
    if (gflags & (GNNCCA_GRAPH_UNSORTED | GNNCCA_GRAPH_IRREGULAR | GNNCCA_GRAPH_BAD_INDEX)) return;   // the second launch does the step
    const int S = p.ell_S, N = p.N;
    const bool pair = S <= 128;                      // two nodes per item
    const int nb = pair ? 1 : (S + 255) / 256;       // 256-slot blocks per node
    const int items = pair ? (N + 1) / 2 : N * nb;
    const int G = gridDim.x, g = blockIdx.x;
    const int i0 = (int)((long long)items * g / G), i1 = (int)((long long)items * (g + 1) / G);
    const int n = i1 - i0;
    if (n <= 0) return;
    constexpr bool nt_store = NT >= 1, nt_load = NT >= 2;
    const long long es = p.e_stride;

    if (wave == 4) {
        // ------------------------------------------------ loader + node epilogue ------------------------------------------------
        __builtin_amdgcn_s_setprio(3);   // one wave feeds four: its instructions go first on the SIMD it shares with consumers
        float wcol[kH];
        float projb_l = 0.f;
        const int o = min(lane, kProjOut - 1);
        if (MSG) {
#pragma unroll
            for (int c = 0; c < kH; ++c) wcol[c] = blob[p.off_projwT + c * kProjOut + o];
            projb_l = blob[p.off_projb + o];
        }
        int cbase = -(1 << 30), segv = 0;   // lane cache of seg_ptr[cbase + lane]
        auto issue = [&](int it) {
            const int I = i0 + it;
            const int node0 = pair ? 2 * I : I / nb;
            const int b = pair ? 0 : I - node0 * nb;
            if (node0 < cbase || node0 + 2 > cbase + 63) {
                cbase = node0;
                segv = p.seg_ptr[min(node0 + lane, N)];
            }
            const int s0 = __builtin_amdgcn_readlane(segv, node0 - cbase);
            float* slot = ring + (it % R) * kRingSlot;
            const int hv = __shfl(segv, min(node0 + min(lane, 2), N) - cbase);
            if (lane < 3) reinterpret_cast<int*>(slot + kRingHdr)[lane] = hv;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_off(slot));
            const long long sl = min((long long)node0 * S + 256 * b + 4 * lane, (long long)N * S - 4);
#pragma unroll
            for (int f = 0; f < kEF; ++f) glds16<nt_load>(p.e + (size_t)f * es + sl, dst + f * 1024);
            const int cb = s0 + 256 * b + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds4(p.col32 + min(cb + 64 * i, p.E - 1), dst + (kRingCol + 64 * i) * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                glds4(p.psq_in + (size_t)min(node0 + j, N - 1) * kPsQStride + min(lane, kPsQStride - 1), dst + (kRingPsq + 64 * j) * 4);
        };
        auto epilogue = [&](int it) {   // the nodes completed by item `it`: combine the partial sums, project, store
            const int I = i0 + it;
            const int node0 = pair ? 2 * I : I / nb;
            const int b = pair ? 0 : I - node0 * nb;
            if (b != nb - 1) return;
            const int* hdr = reinterpret_cast<const int*>(ring + (it % R) * kRingSlot + kRingHdr);
            const float* part = s_part + (it & 1) * 4 * kH;
            const int ch = lane & 31;
            const int nn = pair ? min(2, N - node0) : 1;
            for (int j = 0; j < nn; ++j) {
                const int deg = hdr[j + 1] - hdr[j];
                float v;
                if (pair)
                    v = part[(2 * j) * kH + ch] + part[(2 * j + 1) * kH + ch];
                else
                    v = ((part[ch] + part[kH + ch]) + part[2 * kH + ch]) + part[3 * kH + ch];
                if (p.agg == GNNCCA_AGG_MEAN) v = v / (float)max(deg, 1);
                if (deg == 0) v = 0.f;
                float pr = projb_l;
#pragma unroll
                for (int c = 0; c < kH; ++c) pr = fmaf(wcol[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c)), pr);
                const int node = node0 + j;
                if (lane < kPdStride)
                    p.pd_out[(size_t)node * kPdStride + lane] = pr;
                else if (lane < kProjOut)
                    p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
            }
        };
        RING_T_DECL;
        for (int it = 0; it < R - 1 && it < n; ++it) issue(it);
        RING_T(0);
        if (R - 2 <= n - 1)       // round -1: items 0 and 1 landed before the consumers' first gather
            GNNCCA_WAIT_VM(kRingDma * (R - 3));
        else
            GNNCCA_WAIT_VM(0);
        GNNCCA_RING_BARRIER();
        for (int it = 0; it < n; ++it) {
            // items <= it + 1 must have landed (the consumers read item it + 1's ids this round); issued so far: <= it + R - 2
            if (it + R - 2 <= n - 1)
                GNNCCA_WAIT_VM(kRingDma * (R - 3));
            else
                GNNCCA_WAIT_VM(0);
            RING_T(1);
            GNNCCA_RING_BARRIER();
            RING_T(2);
            if (MSG && it > 0) epilogue(it - 1);      // reads slot (it - 1) % R's header: before that slot is refilled
            RING_T(3);
            if (it + R - 1 < n) issue(it + R - 1);
            RING_T(4);
        }
        GNNCCA_RING_BARRIER();
        if (MSG) epilogue(n - 1);
        RING_T(5);
        RING_T_FLUSH(blockIdx.x + 2048, 0);
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    float bw[3] = {0.f, 0.f, 0.f};
    if (MSG) {
#pragma unroll
        for (int s = 0; s < 3; ++s) bw[s] = blob[p.off_wneb + s * 64 + lane];
    }
    const int half = lane >> 5, ch = lane & 31;
    const int wj = pair ? (wave >> 1) : 0;            // which node of the item this wave works on
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    struct Pd {
        f32x4 a;
        f32x2 b;
    };
    // P_dst rows of this wave's 64 edges of item `it` (its ids and header are in the ring)
    auto gather = [&](int it) {
        const float* slot = ring + (it % R) * kRingSlot;
        const int* hdr = reinterpret_cast<const int*>(slot + kRingHdr);
        const int cpos = pair ? __builtin_amdgcn_readfirstlane(hdr[wj] - hdr[0]) + 64 * (wave & 1) + lane : 64 * wave + lane;
        const int j = (int)min((unsigned)reinterpret_cast<const int*>(slot + kRingCol)[min(cpos, 255)], (unsigned)(N - 1));
        const float* __restrict__ pdj = p.pd_in + (size_t)j * kPdStride;
        Pd r;
        r.a = *reinterpret_cast<const f32x4*>(pdj);
        r.b = *reinterpret_cast<const f32x2*>(pdj + 4);
        return r;
    };
    // One round.  `cur` holds the P_dst rows of item `it` (gathered a round earlier), `nxt` receives item it + 1's: the loop below
    // alternates two register sets, so no in-flight load is ever copied (a copy is a use: the compiler would wait for it, and
    // for every older store, at the end of each round).
    RING_T_DECL;
    auto round = [&](int it, const Pd& cur, Pd& nxt) {
        RING_T(0);
        GNNCCA_RING_BARRIER();
        RING_T(1);
        const int I = i0 + it;
        const int node0 = pair ? 2 * I : I / nb;
        const int b = pair ? 0 : I - node0 * nb;
        const float* slot = ring + (it % R) * kRingSlot;
        const int* hdr = reinterpret_cast<const int*>(slot + kRingHdr);
        const int node = node0 + wj;
        const int seg_s = __builtin_amdgcn_readfirstlane(hdr[wj]);
        const int deg = __builtin_amdgcn_readfirstlane(hdr[wj + 1]) - seg_s;
        const int k0 = pair ? 64 * (wave & 1) : 256 * b + 64 * wave;   // first edge of this wave's chunk inside the node
        const int k = k0 + lane;
        const int pos = pair ? wj * S + 64 * (wave & 1) + lane : 64 * wave + lane;   // slot inside the item
        const bool valid = k < deg;
        float ein[kEF], psrc[kEF];
        const float* psq = slot + kRingPsq + 64 * wj;
#pragma unroll
        for (int f = 0; f < kEF; ++f) ein[f] = slot[f * 256 + min(pos, 255)];
#pragma unroll
        for (int f = 0; f < kEF; ++f) psrc[f] = psq[f];
        const float cinit = MSG ? psq[8 + ch] : 0.f;
        const float pd[kEF] = {cur.a[0], cur.a[1], cur.a[2], cur.a[3], cur.b[0], cur.b[1]};

        float en[kEF];
        f32x2 s2[kEF / 2];
#pragma unroll
        for (int h = 0; h < kEF / 2; ++h) s2[h] = f32x2{psrc[2 * h], psrc[2 * h + 1]} + f32x2{pd[2 * h], pd[2 * h + 1]};
        // `cur` was requested a round ago: the wait above (the compiler's, for every older operation of this wave) is short.
        // Only NOW ask for the next item's rows, so that this round's waits never include them.
        // (the empty statement pins the three sums before it and, through its memory clobber, the gather's loads after it)
        asm volatile("" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]) : : "memory");
        RING_T(2);
        if (it + 1 < n) nxt = gather(it + 1);
        RING_T(3);
#pragma unroll
        for (int gg = 0; gg < kEF; ++gg) {
            const f32x2 x = {ein[gg], ein[gg]};
#pragma unroll
            for (int h = 0; h < kEF / 2; ++h) {
                const f32x2 w = {cw[kFcWee + gg * kEF + 2 * h], cw[kFcWee + gg * kEF + 2 * h + 1]};
                s2[h] = __builtin_elementwise_fma(w, x, s2[h]);
            }
        }
#pragma unroll
        for (int h = 0; h < kEF / 2; ++h) en[2 * h] = fmaxf(s2[h][0], 0.f), en[2 * h + 1] = fmaxf(s2[h][1], 0.f);
        if (p.store_e && valid) {
            const size_t sidx = (size_t)node * S + k;
#pragma unroll
            for (int f = 0; f < kEF; ++f) {
                if (nt_store)
                    __builtin_nontemporal_store(en[f], p.e + (size_t)f * es + sidx);
                else
                    p.e[(size_t)f * es + sidx] = en[f];
            }
        }
        RING_T(4);
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
                    __builtin_nontemporal_store(logit, p.logits + seg_s + k);
                else
                    p.logits[seg_s + k] = logit;
            }
        }
        RING_T(5);
        if (MSG) {
            if (b == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            // one 32-edge tile at a time (16 accumulator registers live instead of 32: the kernel fits 6 waves per SIMD, i.e. four
            // 5-wave workgroups per CU); the two tiles' ReLU'd rows are added to acc one after the other
            float a0[3], a1[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(en[2 * s]), __float_as_uint(en[2 * s + 1]), false, false);
                a0[s] = __uint_as_float(r[0]), a1[s] = __uint_as_float(r[1]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 d;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = cinit;
#pragma unroll
                for (int s = 0; s < 3; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(t ? a1[s] : a0[s], bw[s], d, 0, 0, 0);
                if (k0 + 64 <= deg) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += relu_bits(d[i]);
                } else {
                    const int rem = deg - k0 - 4 * half - 32 * t;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int eo = (i & 3) + 8 * (i >> 2);
                        acc[i] += (eo < rem) ? relu_bits(d[i]) : 0.f;
                    }
                }
            }
            if (b == nb - 1) {
                float v = acc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) v += acc[i];
                v += __shfl_xor(v, 32);
                if (lane < kH) s_part[((it & 1) * 4 + wave) * kH + lane] = v;
            }
        }
    };
    Pd pda, pdb;
    pdb.a = f32x4{0.f, 0.f, 0.f, 0.f};
    pdb.b = f32x2{0.f, 0.f};
    GNNCCA_RING_BARRIER();          // round -1: items 0 and 1 have landed; the loader's matching barrier opens its loop
    pda = gather(0);
    for (int it = 0; it < n; it += 2) {
        round(it, pda, pdb);
        if (it + 1 < n) round(it + 1, pdb, pda);
    }
    RING_T(0);
    GNNCCA_RING_BARRIER();
    RING_T(1);
    RING_T_FLUSH(blockIdx.x, wave);
}

template <bool CLS, bool MSG, int NT>
static hipError_t launch_ring_t(const StepParams& sp, hipStream_t st) {
    static const int ring_r = std::getenv("GNNCCA_RING_R") ? std::atoi(std::getenv("GNNCCA_RING_R")) : 4;
    static const int ring_wpc = std::getenv("GNNCCA_RING_WPC") ? std::atoi(std::getenv("GNNCCA_RING_WPC")) : 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
#define GNNCCA_RING_CASE(RR, WPC)                                                                                            \
    {                                                                                                                        \
        const size_t lds = ((size_t)RR * kRingSlot + 2 * 4 * kH) * sizeof(float);                                            \
        static thread_local int attr_dev = -1;                                                                               \
        if (attr_dev != dev) {                                                                                               \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(mpn_step_ring_kernel<CLS, MSG, NT, RR>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
            if (e != hipSuccess) return e;                                                                                   \
            attr_dev = dev;                                                                                                  \
        }                                                                                                                    \
        const int wpc = ring_wpc > 0 ? ring_wpc : WPC;                                                                       \
        static const bool dbg = std::getenv("GNNCCA_RING_DEBUG") != nullptr;                                                 \
        if (dbg) {                                                                                                           \
            int nblk = -1;                                                                                                   \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, mpn_step_ring_kernel<CLS, MSG, NT, RR>, 320, lds);     \
            std::fprintf(stderr, "ring<%d,%d,%d,%d>: lds %zu B, occupancy API %d blocks/CU, grid %d\n", (int)CLS, (int)MSG, NT, RR, lds, nblk, 256 * wpc); \
        }                                                                                                                    \
        GNNCCA_LAUNCH((mpn_step_ring_kernel<CLS, MSG, NT, RR>), dim3(256 * wpc), dim3(320), lds, st, sp);                    \
    }
    if (ring_r == 6) GNNCCA_RING_CASE(6, 3)
    else if (ring_r == 5) GNNCCA_RING_CASE(5, 3)
    else GNNCCA_RING_CASE(4, 4)
#undef GNNCCA_RING_CASE
    return hipGetLastError();
}

static hipError_t launch_ring(const StepParams& sp, bool msg, hipStream_t st) {
    const int nt = sp.nt_load ? 2 : (sp.nt_store ? 1 : 0);
    const bool cls = sp.cls_layers != 0;
#define GNNCCA_RING_NT(C, M)                                          \
    switch (nt) {                                                     \
        case 2: return launch_ring_t<C, M, 2>(sp, st);                \
        case 1: return launch_ring_t<C, M, 1>(sp, st);                \
        default: return launch_ring_t<C, M, 0>(sp, st);               \
    }
    if (cls && msg) GNNCCA_RING_NT(true, true)
    if (cls && !msg) GNNCCA_RING_NT(true, false)
    if (!cls && msg) GNNCCA_RING_NT(false, true)
    GNNCCA_RING_NT(false, false)
#undef GNNCCA_RING_NT
}

}  // namespace gnncca
