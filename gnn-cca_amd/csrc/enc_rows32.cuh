#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Un-split split-bf16 encoder GEMM with the fused epilogue for MID-SIZE batches (a few thousand ... a few ten thousand
// nodes: the per-GPU share of BASELINE config 4, N = 8192, is the shape it was built for).  There the 256-row form needs
// split-K to fill the chip -- [ks][N][128] fp32 slabs written and re-read (1.65x the algorithmic bytes) plus a tail launch:
// 27.5 + 12 us at N = 8192.  Here a workgroup owns 32 rows x all 128 columns over the whole K, so N / 32 workgroups fill the
// chip un-split and the rest of the encoder runs on the tile while it is on chip.
//   workgroup = 4 waves; wave w owns column tile w: ONE 32 x 32 accumulator, 16 VGPRs;
//   x  : 128 k at a time (512 B of 32 rows = 16 KB, two 32-B pieces per thread), TWO super-chunks ahead in two register sets,
//        split once per element into its three bf16 pieces (the packed form of the 256-row kernel: same roundings) and parked
//        in a double-buffered LDS image [2][3][32 rows][128 k] (48 KB; 16-B granules XOR-swizzled by row);
//   W  : the wave's B fragments are private (nobody else touches its 32 columns), so they go from L2 straight into registers
//        in fragment shape, NST - 1 sub-chunks of 32 k ahead (NST register stages of 24 VGPRs) and never touch LDS;
//   one barrier per 128 k, placed BEFORE the last k-step's MFMAs (its A fragments are already in registers): the first
//   fragment read of the next super-chunk is issued behind the barrier and lands under those six MFMAs.
// Per accumulator the MFMA sequence is the 256-row kernel's (k ascending, six piece products smallest first) and the epilogue
// is enc_finish_32rows: a node's encoder output does not depend on which of the two un-split kernels produced it, bit for bit
// -- a 64-graph shard of config 4 (N = 8192, this kernel) reproduces its graphs' logits inside the 512-graph union
// (N = 65 536, 256-row kernel) exactly.
// The price (measured, profiles/r03_logs/r3_r32_ab1.log, r3_r32_ab3.log, r3_r32_abl1.log, r3_r32_abl2.log, r3_pmc_r32.log): 35 us at N = 4096,
// 41 us at N = 8192 (split-K + tail: 31 / 38), 76-84 us at N = 16 384.  At N <= 8192 there is ONE wave per SIMD and one dependent
// accumulation chain per wave: with every load, LDS access and conversion removed the 768 MFMAs alone take 24 us (timing-only
// ablation) -- a lone wave retires a v_mfma_f32_32x32x16_bf16 per 32 cycles where two / four waves on a SIMD get 24 / 20
// (tools/archive/ubench_mfma_chain.hip), at the ~1.3 GHz the chip sustains under this load -- and nothing hides its own waits: the W
// fragments (every workgroup streams all 1.5 MB of pieces from L2, 515 MB of L2 requests per launch at N = 8192, 84 % hits) cost
// 7.5 us, x and its conversion 3.5 us.  A bare sweep of a 1.5 MB L2-resident table reaches 106-135 GB/s per CU
// (tools/ubench_l2_per_cu.hip: 64 B/clk, the L1's width), this kernel 42-50; reading the pieces in a fragment-ordered 1 KB-contiguous
// pattern instead of 32 rows x 2 x 16 B: 43.5 -> 38.5 us (timing probe); fewer W stages (1-3 sub-chunks ahead): 43-50 us.
// Replaces models/mpn.py:131 (encoder.node_mlp) on such batches.
// ------------------------------------------------------------------------------------------------------------
constexpr int kR32LdsBytes = 2 * 3 * 32 * 128 * 2;   // 49 152

template <bool P3, int NST, int DIST = NST - 1>
__global__ __launch_bounds__(256) void enc_gemm_rows32_fused_kernel(const float* __restrict__ x, const unsigned short* __restrict__ w3, int M, int K,
                                                                    const EncFuseParams fp) {
    constexpr int KA = 128, BK = 32, O = 128;
    static_assert((NST == 2 || NST == 4 || NST == 8) && DIST >= 1 && DIST < NST, "W stages rotate with the four sub-chunks of a super-chunk");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kR32LdsBytes];
    __bf16* sa = reinterpret_cast<__bf16*>(lds_raw);   // [2][3][32][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    if (blockIdx.x == gridDim.x - 1) {   // the plan workgroup: fold the per-block findings, repair if needed
        plan_finish(fp.ei, fp.E, M, fp.seg_ptr, fp.col32, fp.perm, fp.cursor, fp.flags, fp.blockflags, reinterpret_cast<unsigned*>(lds_raw));
        return;
    }
    const int row0 = blockIdx.x * 32;
    const int nsc = K / KA, nchunk = K / BK;   // the host checked: K % 256 == 0
    // x loader: thread (row = tid >> 3, c = tid & 7) owns the 16-B output granules g = c + 8 u (u = 0, 1) of its row: 8 consecutive k
    const int xrow = tid >> 3, xc = tid & 7;
    const float* __restrict__ xsrc = x + (size_t)min(row0 + xrow, M - 1) * K + 8 * xc;
    int adst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) adst[u] = xrow * KA + (((xc + 8 * u) ^ (xrow & 15)) << 3);
    const int afrag = l32 * KA, aswz = l32 & 15;
    const size_t wchunk = (size_t)3 * O * BK;   // bf16 elements per 32-deep k-chunk of w3 ([K/32][3][128][32])
    const unsigned short* __restrict__ wlane = w3 + (size_t)(wave * 32 + l32) * BK + 8 * h;

    f32x4 xs[2][4];          // [register set][2 u + half]: super-chunk c lives in set c & 1
    bf16x8 wb[NST][2][3];    // [stage][k-step][piece]
    bf16x8 af[2][3];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    auto load_x_granule = [&](int c, int u, f32x4 (&dst)[4]) {
        const float* s = xsrc + (size_t)c * KA + 64 * u;
        dst[2 * u] = *reinterpret_cast<const f32x4*>(s);
        dst[2 * u + 1] = *reinterpret_cast<const f32x4*>(s + 4);
    };
    auto load_w = [&](int chunk, bf16x8 (&dst)[2][3]) {
        const unsigned short* src = wlane + (size_t)chunk * wchunk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[ks][p] = *reinterpret_cast<const bf16x8*>(src + (size_t)p * O * BK + ks * 16);
    };
    // one granule (8 consecutive k of one row) -> three bf16x8 pieces -> LDS.  v_cvt_pk_bf16_f32 rounds a pair, the pair's float
    // images are one shift and one mask of that word, the remainders one packed subtract (exact) -- enc_gemm_split_lds_kernel's form
    auto convert_store = [&](int stage, int u, const f32x4 (&src)[4]) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned w[3][4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            f32x2 v = {src[2 * u + (pr >> 1)][2 * (pr & 1)], src[2 * u + (pr >> 1)][2 * (pr & 1) + 1]};
#pragma unroll
            for (int lev = 0; lev < 3; ++lev) {
                const unsigned word = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                w[lev][pr] = word;
                if (lev < 2)
                    v = __builtin_elementwise_fma(f32x2{-1.f, -1.f}, f32x2{__uint_as_float(word << 16), __uint_as_float(word & 0xffff0000u)}, v);
            }
        }
        __bf16* a = sa + (size_t)stage * 3 * 32 * KA + adst[u];
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(a + p * 32 * KA) = u32x4{w[p][0], w[p][1], w[p][2], w[p][3]};
    };
    auto read_a = [&](int stage, int s8, bf16x8 (&dst)[3]) {   // k-step s8 of the super-chunk: granule 2 s8 + h of row l32
        const __bf16* a = sa + (size_t)stage * 3 * 32 * KA + afrag + (((2 * s8 + h) ^ aswz) << 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(a + p * 32 * KA);
    };
    auto mfma6 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
        // smallest terms first -- the order of every split-bf16 GEMM in this file
        if (!P3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    };
    // one super-chunk c of parity PAR: MFMAs from A stage PAR; x(c + 1) (register set PAR ^ 1) -> A stage PAR ^ 1; x(c + 3) requested
    // into the freed registers; W sub-chunk 4 c + j from stage (4 c + j) % NST, sub-chunk 4 c + j + NST - 1 requested.
    auto body = [&](int c, auto par) {
        constexpr int PAR = decltype(par)::value;
        constexpr int WS0 = (4 * PAR) % NST;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int j = s >> 1;
            if ((s & 1) == 0) load_w(min(4 * c + j + DIST, nchunk - 1), wb[(WS0 + j + DIST) % NST]);
            if (s == 7) {
                __syncthreads();   // every wave holds its last fragments of stage PAR; the image of x(c + 1) is complete
                read_a(PAR ^ 1, 0, af[0]);
            } else {
                read_a(PAR, s + 1, af[(s + 1) & 1]);
            }
            mfma6(af[s & 1], wb[(WS0 + j) % NST][s & 1]);
            if (s == 1 || s == 4) {
                const int u = s == 1 ? 0 : 1;
                convert_store(PAR ^ 1, u, xs[PAR ^ 1]);
                load_x_granule(min(c + 3, nsc - 1), u, xs[PAR ^ 1]);
            }
        }
    };
    // prologue: x(0) -> A stage 0; x(1), x(2) in flight; W sub-chunks 0 .. NST - 2 in flight
    // (issue order matters: s_waitcnt vmcnt counts in issue order and hipcc takes, at the loop head, the more conservative of the
    // prologue's and the steady state's counts -- x first, then the W stages, as in the loop where an x granule is two super-chunks old)
    f32x4 x0[4];
    load_x_granule(0, 0, x0);
    load_x_granule(0, 1, x0);
    load_x_granule(min(1, nsc - 1), 0, xs[1]);
    load_x_granule(min(1, nsc - 1), 1, xs[1]);
    load_x_granule(min(2, nsc - 1), 0, xs[0]);
    load_x_granule(min(2, nsc - 1), 1, xs[0]);
#pragma unroll
    for (int q = 0; q < DIST; ++q) load_w(min(q, nchunk - 1), wb[q]);
    convert_store(0, 0, x0);
    convert_store(0, 1, x0);
    __syncthreads();
    read_a(0, 0, af[0]);
    for (int c = 0; c < nsc; c += 2) {
        body(c, std::integral_constant<int, 0>{});
        body(c + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();   // the stages are dead: the epilogue's tiles go over them
    // ---- fused epilogue: h1 = [ReLU](acc + b1) -> LDS -> enc_finish_32rows ------------------------------------------------
    float* H1 = reinterpret_cast<float*>(lds_raw);            // [32][132]
    float* Dp = H1 + 32 * kFinLD1;                            // [4][32][33]
    float* H0 = Dp + 4 * 32 * kFinLDP;                        // [32][36]
    {
        const int col = wave * 32 + l32;
        const float bias = fp.b1[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
            const float v = acc[i] + bias;
            H1[rl * kFinLD1 + col] = fp.relu_prev ? fmaxf(v, 0.f) : v;
        }
    }
    __syncthreads();
    EncFinishOut fo;
    fo.W2rm = fp.W2, fo.b2 = fp.b2, fo.projwT = fp.projwT, fo.projb = fp.projb;
    fo.h0 = fp.h0, fo.trace_h = fp.trace_h, fo.pd_out = fp.pd_out, fo.psq_out = fp.psq_out;
    enc_finish_32rows(H1, Dp, H0, fo, enc_finish_preload(fo), row0, M);
}

}  // namespace gnncca
