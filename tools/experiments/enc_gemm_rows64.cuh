// Shelved experiment (round 2): see DESIGN.md, round-2 experiment log.  Not compiled.
// ------------------------------------------------------------------------------------------------------------
// Structure 3 ("64 rows x 512 B"): the split-bf16 GEMM rebuilt around how x streams.  Ablations of the 256-row kernel
// above showed it bound by its x stream (160 us at N = 65 536 with every MFMA removed), and tools/ubench_rowtile.hip
// showed what that stream is worth per tile shape in a bare copy loop: 256 rows x 128 B per visit 4.4 TB/s, 64 rows x
// 512 B per visit 5.4-5.6 TB/s (a DRAM page is visited for four lines instead of one) -- so:
//   workgroup = 4 waves = 64 rows x all 128 columns; x arrives 128 k (512 B per row, 8 x 16 B per thread) at a time, a
//   whole A chunk ahead in registers, is converted once per element to its three bf16 pieces and parked in LDS in the
//   fragment image (48 KB, single stage, 16-B granules XOR-swizzled by row: ds_read_b128 of any 16 rows hits 64 banks);
//   wave w owns column tile w for both row tiles, so its W fragments are private: they come straight from L2 into registers
//   (fragment-shaped 16-B loads from the chunk-contiguous blob, one 32-k sub-chunk ahead) and never touch LDS -- two
//   barriers per 128 k instead of eight (a first version that staged W through a single LDS stage: 285 us at N = 65 536).
//   64 KB of LDS => TWO workgroups per CU, unsynchronised: while one converts or waits at a barrier the other feeds the
//   matrix pipe -- the overlap the stagger experiment tried to force inside one workgroup comes for free.
//   The price: W is re-read from L2 once per 64 rows (1.5 GB at N = 65 536, 4x the 256-row kernel's).
// FUSE: as above (rest of the encoder + projections + plan fold in the epilogue) when the launch is un-split.
// ------------------------------------------------------------------------------------------------------------
constexpr size_t kR64LdsBytes = (size_t)3 * 64 * 128 * 2 + 16384;   // A image 49 152 B (+ room for the epilogue's tiles): 2 workgroups / CU
template <bool FUSE, bool P3>
__global__ __launch_bounds__(256, 2) void enc_gemm_rows64_kernel(const float* __restrict__ x, const unsigned short* __restrict__ w3,
                                                                 float* __restrict__ out, int M, int K, int O, int kslice,
                                                                 const EncFuseParams fp) {
    constexpr int RA = 64, KA = 128, BK = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* sa = reinterpret_cast<__bf16*>(lds_raw);   // [3][64][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    if (FUSE && blockIdx.x == gridDim.x - 1) {   // the plan workgroup
        plan_finish(fp.ei, fp.E, M, fp.seg_ptr, fp.col32, fp.perm, fp.cursor, fp.flags, fp.blockflags,
                    reinterpret_cast<unsigned*>(lds_raw));
        return;
    }
    const int row0 = blockIdx.x * RA;
    const int kbeg = blockIdx.y * kslice;
    const int nchunk = min(kslice, K - kbeg) / KA;
    // x loader: thread (r0 = tid >> 5, c = tid & 31) takes float4 c of rows r0 + 8 u: a wave instruction = 2 rows x 512 B
    const int r0 = tid >> 5, c4 = tid & 31;
    int adst[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = r0 + 8 * u;
        adst[u] = row * KA + (((c4 >> 1) ^ (row & 15)) << 3) + ((c4 & 1) << 2);
    }
    // W: wave w owns column tile w for both row tiles, so its B fragments are private -- they come straight from the
    // chunk-contiguous blob ([K/32][3][128][32] bf16) into registers, one 32-k sub-chunk ahead, and never touch LDS:
    // lane (n, h) reads the 16-B granule 2 ks + h of row n (the two halves read adjacent granules: 32 B per row and load)
    const size_t wchunk = (size_t)3 * O * BK;
    const unsigned short* wlane = w3 + (size_t)(wave * 32 + l32) * BK + h * 8;
    f32x4 xr[8];
    bf16x8 wb[2][2][3];   // [stage][ks][piece]
    auto load_x = [&](int c) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            xr[u] = *reinterpret_cast<const f32x4*>(x + (size_t)min(row0 + r0 + 8 * u, M - 1) * K + kbeg + c * KA + c4 * 4);
    };
    auto load_w = [&](int gs, bf16x8 (&dst)[2][3]) {   // gs: 32-k sub-chunk index inside this workgroup's k range
        const unsigned short* src = wlane + (size_t)(kbeg / BK + gs) * wchunk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[ks][p] = *reinterpret_cast<const bf16x8*>(src + (size_t)p * O * BK + ks * 16);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bf16x4 p0, p1, p2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = xr[u][q];
                const __bf16 h0 = (__bf16)v;
                const float r1 = v - (float)h0;
                const __bf16 h1 = (__bf16)r1;
                const float r2 = r1 - (float)h1;
                p0[q] = h0, p1[q] = h1, p2[q] = (__bf16)r2;
            }
            *reinterpret_cast<bf16x4*>(sa + adst[u]) = p0;
            *reinterpret_cast<bf16x4*>(sa + RA * KA + adst[u]) = p1;
            *reinterpret_cast<bf16x4*>(sa + 2 * RA * KA + adst[u]) = p2;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
    int arow[2], aswz[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) arow[rt] = (rt * 32 + l32) * KA, aswz[rt] = (rt * 32 + l32) & 15;
    auto mfma_sub = [&](int sub, const bf16x8 (&bf)[2][3]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ga = sub * 4 + ks * 2 + h;   // 16-B granule inside the A row (16 per row)
            bf16x8 af[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    af[rt][p] = *reinterpret_cast<const bf16x8*>(sa + p * RA * KA + arow[rt] + ((ga ^ aswz[rt]) << 3));
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                // smallest terms first
                if (!P3) {   // the three 2^-16-order terms (GNNCCA_OPT_ENC_SPLIT3 drops them)
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][2], bf[ks][0], acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[ks][1], acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ks][2], acc[rt], 0, 0, 0);
                }
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[ks][0], acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ks][1], acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ks][0], acc[rt], 0, 0, 0);
            }
        }
    };
    load_x(0);
    load_w(0, wb[0]);
    const int nsub = nchunk * 4;
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0) __syncthreads();            // every wave is done reading the A image of chunk c - 1
        store_a();
        if (c + 1 < nchunk) load_x(c + 1);     // in flight during the four sub-chunks below
        __syncthreads();                       // the A image of chunk c is visible
        // four 32-k sub-chunks, no barrier: B is private to the wave (register double buffer, static indices)
        if (c * 4 + 1 < nsub) load_w(c * 4 + 1, wb[1]);
        mfma_sub(0, wb[0]);
        if (c * 4 + 2 < nsub) load_w(c * 4 + 2, wb[0]);
        mfma_sub(1, wb[1]);
        if (c * 4 + 3 < nsub) load_w(c * 4 + 3, wb[1]);
        mfma_sub(2, wb[0]);
        if (c * 4 + 4 < nsub) load_w(c * 4 + 4, wb[0]);
        mfma_sub(3, wb[1]);
    }
    if (!FUSE) {
        float* __restrict__ dst = out + (size_t)blockIdx.y * M * O;
        const int col = wave * 32 + l32;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M) dst[(size_t)row * O + col] = acc[rt][i];
            }
        return;
    }
    // ---- fused epilogue (O == 128, un-split), every MFMA a v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains) -----------------------
    constexpr int LD1 = 132, LDP = 33, LD0 = 36;
    float* H1 = reinterpret_cast<float*>(lds_raw);      // [64][132]
    float* Dp = H1 + 64 * LD1;                          // [4][32][33]  partial tiles of layer 2
    float* H0 = Dp + 4 * 32 * LDP;                      // [64][36]
    __syncthreads();                                    // the last MFMAs have read A / B
    {
        const int col = wave * 32 + l32;
        const float bias = fp.b1[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = acc[rt][i] + bias;
                H1[rl * LD1 + col] = fp.relu_prev ? fmaxf(v, 0.f) : v;
            }
    }
    __syncthreads();
    {   // layer 2: wave -> (row tile wave >> 1, k half wave & 1); lane (m, h) feeds k = 64 kh + 32 h + s at step s
        const int rt = wave >> 1, kh = wave & 1;
        const float* hr = H1 + (rt * 32 + l32) * LD1 + 64 * kh + 32 * h;
        const float* wr = fp.W2 + (size_t)l32 * 128 + 64 * kh + 32 * h;
        float av[32], bv[32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(wr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[4 * j + q] = a4[q], bv[4 * j + q] = b4[q];
        }
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) Dp[(wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * LDP + l32] = d[i];
    }
    __syncthreads();
    {
        const int r = tid >> 2, c8 = (tid & 3) * 8;     // 64 rows x 32 channels, 8 per thread
        const int row = row0 + r;
        const float* d0 = Dp + ((2 * (r >> 5)) * 32 + (r & 31)) * LDP;
        const float* d1 = d0 + 32 * LDP;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = c8 + q;
            const float v = fmaxf((d0[c] + d1[c]) + fp.b2[c], 0.f);
            H0[r * LD0 + c] = v;
            if (row < M) {
                fp.h0[(size_t)row * kH + c] = v;
                if (fp.trace_h) fp.trace_h[(size_t)row * kH + c] = v;
            }
        }
    }
    __syncthreads();
    {   // projections: wave -> (row tile wave >> 1, slot tile wave & 1); lane (m, h) feeds k = 16 h + s
        const int rt = wave >> 1, slot = 32 * (wave & 1) + l32;
        const bool on = slot < kProjOut;
        float a2[16], b2v[16];
        const float* hr = H0 + (rt * 32 + l32) * LD0 + 16 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) a2[4 * j + q] = a4[q];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) b2v[s] = on ? fp.projwT[(16 * h + s) * kProjOut + slot] : 0.f;
        f32x16 pacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b2v[s], pacc, 0, 0, 0);
        const float pb = on ? fp.projb[slot] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (row < M && on) {
                const float v = pacc[i] + pb;
                if (slot < kPdStride)
                    fp.pd_out[(size_t)row * kPdStride + slot] = v;
                else
                    fp.psq_out[(size_t)row * kPsQStride + slot - kPdStride] = v;
            }
        }
    }
}

