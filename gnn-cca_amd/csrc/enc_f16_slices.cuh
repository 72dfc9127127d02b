#pragma once
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// First encoder layer on graphs and batches of up to a few thousand nodes, round 6: part[ks][M][128] = x[:, slice ks] . W1[:, slice ks]^T in the
// fp16-split arithmetic of enc_f16.cuh (x = x0 + x1 / 2048, w = w0 + w1 / 2048; three v_mfma_f32_32x32x16_f16 products in two accumulators;
// the weights' fragment-major fp16 image BlobHeader::enc_w2h), 32-ROW tiles, K split over the grid AND over the two wave groups of a workgroup.
// Replaces the first nn.Linear of encoder.node_mlp (models/mpn.py:131 <- models/mlp.py:13) where enc_gemm_plan_kernel (N < 384: exact-fp32
// MFMA, 32 slabs) and enc_gemm_split_direct_kernel (384 <= N: six bf16 products, 128-row workgroups, up to 16 slabs) ran until round 5.
//
// These sizes are LATENCY: a dense256 graph's encoder is 3 MB of operands and 6 us of launch, so the kernel is built to pay ONE memory round
// trip per wave.  Workgroup (row tile rt, slice ks) owns 32 rows x 128 columns x Ks <= 256 of k; eight waves = 4 column tiles x 2 k-halves.
//   A: the tile's x[32 rows][Ks] goes into LDS by LDS-DMA at the kernel's start -- every 128-byte line once, whole, no VGPR staging; 32-deep
//      chunks of [32 rows][128 B] with the 16-byte granules XOR-swizzled on the SOURCE address (enc_f16.cuh's layout: conflict-free
//      ds_read_b128 of the A fragments); the waves split the fragments into the two fp16 pieces in registers;
//   B: wave (ct, kh) requests ALL fragments of its k-half and column tile up front (<= 16 loads of 16 bytes per lane, straight from the
//      L2-resident fragment-major image) right behind its share of the DMA, so both operands fly in the same round trip;
//   then <= 8 k-steps of three MFMAs, the two k-halves meet in 16 KB of LDS (k-half 0 + k-half 1), waves 0-3 store the [32][128] tile of
//   slab ks.  The existing tail kernels (enc_tail_fast_kernel / enc_tail_mfma_kernel) sum the slabs in slab order and finish the encoder; the
//   graph plan rides as extra workgroups (two 256-thread plan blocks side by side in a 512-thread workgroup), as in the kernels replaced.
// (First version of the round: four waves per workgroup, operands straight from global memory into double-buffered registers, 140 VGPRs --
// two groups of 32 k in flight made dense1024 a chain of four round trips per wave, 9.2 us, and the four column-tile waves read every x
// line four times; docs/experiments_r06.md.)
// RANGE.  fp16's exponent is narrow: a wave that meets a finite |x| >= 65520 in its operands, a wave whose largest |x| is below 2^-10 (kF16Tiny: the
// pieces' 2^-36 absolute precision would show), or any wave when a weight is beyond fp16 (the packers' flag words), recomputes its own 32 x 32 partial tile as an exact-fp32 MFMA chain (v_mfma_f32_32x32x2_f32, the arithmetic of
// enc_gemm_plan_kernel) -- no input makes this kernel wrong, unusual ones make it slower.  (Per WAVE, before the k-halves meet.)
// ------------------------------------------------------------------------------------------------------------
constexpr int kF16SlThreads = 512;
constexpr int kF16SlMaxKs = 256;     // k per workgroup: the A tile is 32 rows x Ks floats = 32 KB of LDS at most

struct EncF16SlicesParams {
    const float* x;
    const unsigned short* w2h;     // BlobHeader::enc_w2h
    const unsigned* w_bad;         // BlobHeader::enc_w2h_bad
    const float* w32;              // the same weight in fp32, row-major [128][K] (the range arm)
    float* part;                   // [nks][M][128]
    int M, K, nrt, nks, Ks;        // Ks = K / nks: a multiple of 64, <= kF16SlMaxKs
    int force_arm;                 // diagnostics / tests: every wave takes the fp32 arm
};

// STEPS = 16-deep k-steps per wave = Ks / 32: 8 (Ks = 256), 4 (Ks = 128), 2 (Ks = 64)
template <int STEPS>
__device__ __forceinline__ void enc_gemm_f16_slices_body(const EncF16SlicesParams p, const EncPlanParams plan) {
    __shared__ __attribute__((aligned(16))) unsigned char s_a[kF16SlMaxKs / 32 * 4096];   // the A tile: Ks / 32 chunks of [32 rows][128 B]
    __shared__ __attribute__((aligned(16))) float s_half[4 * 16 * 64];                   // k-half 1's partial tiles, accumulator layout
    __shared__ unsigned s_plan_fl[2];
    const int tid = threadIdx.x;
    const int gemm_blocks = p.nrt * p.nks;
    if ((int)blockIdx.x >= gemm_blocks) {   // the graph plan: TWO 256-thread plan blocks side by side per workgroup
        plan_block(2 * ((int)blockIdx.x - gemm_blocks) + (tid >> 8), plan.ei, plan.E, plan.N, plan.seg_ptr, plan.col32, plan.blockflags, &s_plan_fl[tid >> 8],
                   plan.ell_S, plan.plan_span, tid & 255);
        return;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int ct = wave & 3, kh = wave >> 2;
    const int rt = (int)blockIdx.x % p.nrt, ks = (int)blockIdx.x / p.nrt;   // neighbours in the grid share a slice of W
    const int M = p.M, K = p.K;
    const int row0 = rt * 32;
    constexpr int Kw = 16 * STEPS;                  // this wave's k range: [kbeg, kbeg + Kw)
    const int kslice = ks * (2 * Kw);
    const int kbeg = kslice + kh * Kw;
    // ---- A: the tile's 2 * Kw / 32 chunks by LDS-DMA, four 1 KB instructions (8 rows x 128 B) per chunk, dealt to the eight waves ------------
    {
        const int rows_here = min(32, M - row0);
        const rsrc_t rx = make_rsrc(p.x + (size_t)row0 * K, (unsigned long long)rows_here * K * 4);   // rows beyond M read as zero
        constexpr int NI = (2 * Kw / 32) * 4;        // DMA instructions of the tile: 32 / 16 / 8
#pragma unroll
        for (int u = 0; u < (NI + 7) / 8; ++u) {
            const int i = wave + 8 * u;              // instruction i = (chunk i / 4, row group i % 4)
            if (NI >= 8 || i < NI) {
                const int c = i >> 2, j = i & 3;
                const int rl = 8 * j + (lane >> 3);
                const int gs = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);      // source granule that lands in LDS granule slot (lane & 7) of row rl
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_dma_ptr)(s_a + c * 4096 + j * 1024), 16, (unsigned)((size_t)rl * K * 4 + gs * 16),
                                                         (unsigned)(kslice + 32 * c) * 4u, 0, 0);
            }
        }
    }
    // ---- B: every fragment of this wave's k-half and column tile, requested right behind the DMA ---------------------------------------------
    const unsigned char* __restrict__ wimg = reinterpret_cast<const unsigned char*>(p.w2h) + (size_t)(kbeg >> 5) * kF16WSlot + (size_t)lane * 16 + (size_t)ct * 2048;
    f16x8 wb[STEPS][2];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        // k-step s of the wave = half (s & 1) of chunk kbeg / 32 + s / 2 (kbeg is a multiple of 32)
        const size_t off = (size_t)(s >> 1) * kF16WSlot + (size_t)(s & 1) * 1024;
        wb[s][0] = *reinterpret_cast<const f16x8*>(wimg + off);
        wb[s][1] = *reinterpret_cast<const f16x8*>(wimg + off + 8192);
    }
    const unsigned wbad = p.w_bad[lane];            // weights beyond fp16: one flag word per lane (kW2hBadWords = 64)
    // the tile has landed once every wave's DMA has.  (vmcnt(0): the B fragments come from L2 and are needed two instructions later anyway; a
    // counted wait would have to trust that the scheduler keeps every ordinary load behind the DMA instructions.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] = 0.f, accB[i] = 0.f;
    float amax = 0.f;
    const int aswz = (l32 >> 1) & 7;
    const unsigned char* abase = s_a + (size_t)(kh * Kw / 32) * 4096 + l32 * 128;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const unsigned char* a = abase + (s >> 1) * 4096;
        const int sc = s & 1;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(a + (((4 * sc + 2 * h) ^ aswz) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(a + (((4 * sc + 2 * h + 1) ^ aswz) << 4));
        f16x8 a0, a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 v = q < 2 ? f32x2{v0[2 * q], v0[2 * q + 1]} : f32x2{v1[2 * q - 4], v1[2 * q - 3]};
            const f16x2 p0 = __builtin_convertvector(v, f16x2);                       // v_cvt_pk_f16_f32: round to nearest even
            const f32x2 r = (v - f32x2{(float)p0[0], (float)p0[1]}) * 2048.0f;       // exact
            const f16x2 p1 = __builtin_convertvector(r, f16x2);
            a0[2 * q] = p0[0], a0[2 * q + 1] = p0[1];
            a1[2 * q] = p1[0], a1[2 * q + 1] = p1[1];
            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
        }
        accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wb[s][0], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wb[s][1], accB, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wb[s][0], accB, 0, 0, 0);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(accB[i], 1.0f / 2048.0f, accA[i]);
    // ---- the range arm, per wave: this wave's 32 x 32 partial tile again as an exact fp32 FMA chain (k-permuted operands as in gemm_tile:
    //      lane (r, h) feeds k = kc + 8 h + s at MFMA step s to both operands; 16 deep at a time) ---------------------------------------------
    // ... or whose largest |x| is below 2^-10 without being zero (the pieces' absolute precision would show as relative error)
    const bool too_small = __builtin_amdgcn_ballot_w64(amax >= kF16Tiny) == 0ull && __builtin_amdgcn_ballot_w64(amax > 0.f) != 0ull;
    const bool out_of_range = too_small || __builtin_amdgcn_ballot_w64(!(amax < kF16Limit) || wbad != 0u) != 0ull;
    if (out_of_range || p.force_arm) {
        const float* __restrict__ ap = p.x + (size_t)min(row0 + l32, M - 1) * K;
        const float* __restrict__ wp = p.w32 + (size_t)(ct * 32 + l32) * K;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int kc = kbeg; kc < kbeg + Kw; kc += 16) {
            float a[8], b[8];
            const int k0 = kc + 8 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k0 + 4 * j);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wp + k0 + 4 * j);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[4 * j + q] = av[q], b[4 * j + q] = bv[q];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
    }
    // ---- k-half 1 hands its tile to k-half 0 through LDS; waves 0-3 store slab ks ------------------------------------------------------------
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_half[(ct * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kh == 1) return;
    float* __restrict__ dst = p.part + ((size_t)ks * M + row0) * 128 + ct * 32 + l32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row0 + rl < M) dst[(size_t)rl * 128] = acc[i] + s_half[(ct * 16 + i) * 64 + lane];
    }
}

// (concrete kernels around the body: with this toolchain a __global__ TEMPLATE that issues the LDS-DMA builtin was not emitted, enc_f16.cuh)
__global__ __launch_bounds__(kF16SlThreads) void enc_gemm_f16_slices8_kernel(const EncF16SlicesParams p, const EncPlanParams plan) { enc_gemm_f16_slices_body<8>(p, plan); }
__global__ __launch_bounds__(kF16SlThreads) void enc_gemm_f16_slices4_kernel(const EncF16SlicesParams p, const EncPlanParams plan) { enc_gemm_f16_slices_body<4>(p, plan); }
__global__ __launch_bounds__(kF16SlThreads) void enc_gemm_f16_slices2_kernel(const EncF16SlicesParams p, const EncPlanParams plan) { enc_gemm_f16_slices_body<2>(p, plan); }

}  // namespace gnncca
