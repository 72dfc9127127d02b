#pragma once
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// First encoder layer on big batches, round 5: the "fp16-split" GEMM  out = x . W^T  (K = in, O = 128; this kernel from 8193 nodes on, the
// 32-row kernel at the end of the file for 4096 ... 8192).
// Replaces the first nn.Linear of encoder.node_mlp (models/mpn.py:131 <- models/mlp.py:13) and, un-split (FUSE), the rest of the encoder.
//
// ARITHMETIC.  x = x0 + x1 / 2048 and w = w0 + w1 / 2048 with fp16 pieces: x0 = fp16(x), x1 = fp16((x - x0) * 2048) -- the residual is exact
// in fp32 and, scaled by 2^11, sits in x0's exponent range, so the pair carries 22 significant bits of x (error <= 2^-22 |x|, or 2^-36
// absolute below fp16's normal range).  x . w ~= x0 w0 + (x0 w1 + x1 w0) / 2048: THREE piece products on v_mfma_f32_32x32x16_f16 (every
// product of two fp16 values is exact in fp32, accumulation is fp32) in two accumulators (the unscaled and the 2^-11 one), against the six
// bf16 products of round 2's form -- half the matrix work for the same class of accuracy (measured against an fp64 evaluation the result
// is closer than an fp32 GEMM's: tools/time_encoder.py --check, tests/test_gpu_parity.py).  fp16 has a narrow exponent: a workgroup whose x
// holds a finite magnitude >= 65520 (it would round to infinity), or any workgroup when a WEIGHT does (flag words written by the packers),
// recomputes its tile on the bf16 six-product arm inside the same launch (`bf16_arm`: range of fp32, round 2's arithmetic); so does (round 6) a
// workgroup whose LARGEST |x| is below 2^-10 without being zero (kF16Tiny: the pieces' 2^-36 absolute precision would show as relative error against
// an fp32 GEMM) -- no input makes this kernel wrong, unusual ones make it slower.
//
// DATA MOVEMENT.  256 rows x 128 columns per workgroup, 8 waves; wave w owns rows [32 w, 32 w + 32) and ALL 128 columns, so the x operand
// is wave-private: every wave streams its own 32 rows by LDS-DMA (buffer_load_dwordx4 ... lds: 8 rows x 128 B per instruction, full
// lines, no VGPR staging, no ds_write) into its own three-slot ring of raw fp32, two chunks ahead, and needs NO barrier for it -- only its
// own counted vmcnt.  The A fragments are read back as fp32 (lane (r, h): 32 B of row r per 16-deep k-step, granules XOR-swizzled on the
// SOURCE address so that ds_read_b128 is conflict-free) and split into the two fp16 pieces in registers: each element is converted once.
// W (pre-split at pack time, BlobHeader::enc_w2h: fragment-major, a 16 KB image per 32-deep chunk) goes through a shared three-slot ring the
// same way, ONE workgroup barrier per 32-deep chunk; a chunk's six DMA instructions per wave are issued one at a time between the MFMA groups
// of the chunk before (a burst after the barrier stalled half the waves in ISSUE for 1 700 cycles per chunk); from 128 MB of x on the x
// requests carry the non-temporal policy (mpn_forward.hip).  LDS traffic per chunk and CU: 48 KB written by DMA + 160 KB of fragment reads, against 72 KB of
// ds_write + 192 KB of reads in the 256-row bf16 kernel (encoder.cuh), and no conversion-store pass.
// (Option, off by default: workgroup b walks its k chunks from chunk (37 b) mod nk on -- see enc_gemm_split_lds_kernel and mpn_forward.hip.)
// LDS: x rings 8 x 3 x 4 KB + W ring 3 x 16 KB = 144 KB (= kLdsGemmBytes); the fused epilogue reuses it.
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_dma_ptr;

constexpr int kF16Stages = 3;
constexpr int kF16XSlot = 32 * 128;                                   // bytes per wave and stage: 32 rows x 32 floats
constexpr int kF16WSlot = 2 * 128 * 64;                               // bytes per stage: 2 pieces x 128 columns x 32 halfs
constexpr int kF16XBytes = 8 * kF16Stages * kF16XSlot;                // 98 304
constexpr size_t kF16LdsBytes = (size_t)kF16XBytes + (size_t)kF16Stages * kF16WSlot;   // 147 456

// The rest of encoder.node_mlp on a 256 x 128 tile h1 that sits in LDS as H1[256][132] (bias and ReLU applied): wave w finishes rows
// [32 w, 32 w + 32) -- layer 2 (four partial 32x32x2 f32 MFMA tiles over k quarters, summed ((d0 + d1) + d2) + d3), h0, the step-1
// projections.  The arithmetic of enc_gemm_split_lds_kernel's fused epilogue (encoder.cuh), statement for statement.
__device__ __forceinline__ void enc_finish_tile_rows(float* H1, int wave, int l32, int h, int row0, int M, const EncFuseParams& fp) {
    constexpr int LD1 = 132, LD0 = 36;
    float* hblk = H1 + (size_t)wave * 32 * LD1;
    f32x16 d;
    {
        f32x16 dq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* hrow = hblk + l32 * LD1 + 32 * q + 16 * h;
            const float* w2row = fp.W2 + (size_t)l32 * 128 + 32 * q + 16 * h;
            float av[16], bv[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(hrow + 4 * j);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(w2row + 4 * j);
#pragma unroll
                for (int t = 0; t < 4; ++t) av[4 * j + t] = a4[t], bv[4 * j + t] = b4[t];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[q][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) dq[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], dq[q], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = ((dq[0][i] + dq[1][i]) + dq[2][i]) + dq[3][i];
    }
    const float bias2 = fp.b2[l32];
    float* H0 = hblk;   // [32][36], over this wave's own (consumed) rows of H1
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
        const float v = fmaxf(d[i] + bias2, 0.f);
        H0[rl * LD0 + l32] = v;
        const int row = row0 + wave * 32 + rl;
        if (row < M) {
            fp.h0[(size_t)row * kH + l32] = v;
            if (fp.trace_h) fp.trace_h[(size_t)row * kH + l32] = v;
        }
    }
    float a2[16];
    {
        const float* hr = H0 + l32 * LD0 + 16 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) a2[4 * j + q] = a4[q];
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int slot = 32 * t + l32;
        const bool on = slot < kProjOut;
        float b2v[16];
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
            const int row = row0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
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

// The bf16 six-product arm for a tile the fp16 pieces cannot carry: round 2's arithmetic (encoder.cuh: enc_gemm_split_direct_kernel's
// loop -- x straight from global memory into three bf16 fragments, the chunk's W pieces through two LDS stages, one barrier per chunk) in
// this kernel's tiling (wave = 32 rows x 128 columns).  Rare by construction; written for correctness, not speed.
// `active`: false for waves beyond the eight that compute (the 32-row kernel's loader waves): they only keep the barriers company.
// NC / cbeg: the wave computes the NC 32-column tiles from tile cbeg on (the 32-row kernel takes its tile in two halves: with all four
// accumulator tiles next to that kernel's prefetched epilogue operands the arm spilled registers -- and a kernel with ANY scratch starts
// its waves more slowly, arm or no arm).
template <int NC = 4>
__device__ __forceinline__ void enc_f16_bf16_arm(f32x16 (&acc)[NC], const float* __restrict__ x, const unsigned short* __restrict__ w3, int M, int K,
                                                 int row0w, int kbeg, int nk, unsigned char* lds_raw, bool active = true, int cbeg = 0) {
    constexpr int BN = 128, BK = 32, LDK = 40;
    typedef __bf16 (*wsm_t)[3][BN][LDK];
    wsm_t wsm = reinterpret_cast<wsm_t>(lds_raw);   // [2][3][128][40] bf16 = 61 440 B
    const int tid = active ? threadIdx.x : 0, lane = tid & 63, h = lane >> 5;
    const size_t plane = (size_t)BN * BK;
    const float* __restrict__ xrow = x + (size_t)min(row0w + (lane & 31), M - 1) * K + kbeg + 8 * h;
    int wp[3], wcol[3], wk[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int idx = tid + 512 * u;
        wp[u] = idx >> 9, wcol[u] = (idx & 511) >> 2, wk[u] = (idx & 3) * 8;
    }
    f32x4 xreg[4];
    bf16x8 wreg[3], afrag[2][3];
    auto load_next = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xreg[2 * ks] = *reinterpret_cast<const f32x4*>(xrow + kt * BK + ks * 16);
            xreg[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xrow + kt * BK + ks * 16 + 4);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            wreg[u] = *reinterpret_cast<const bf16x8*>(w3 + ((size_t)(kbeg / BK + kt) * 3 + wp[u]) * plane + wcol[u] * BK + wk[u]);
    };
    auto convert_x = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = xreg[2 * ks + (q >> 2)][q & 3];
                const __bf16 h0 = (__bf16)v;
                const float r1 = v - (float)h0;
                const __bf16 h1 = (__bf16)r1;
                const float r2 = r1 - (float)h1;
                afrag[ks][0][q] = h0, afrag[ks][1][q] = h1, afrag[ks][2][q] = (__bf16)r2;
            }
    };
    auto store_w = [&](int stage) {
        if (active)
#pragma unroll
            for (int u = 0; u < 3; ++u) *reinterpret_cast<bf16x8*>(&wsm[stage][wp[u]][wcol[u]][wk[u]]) = wreg[u];
    };
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    load_next(0);
    store_w(0);
    convert_x();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) load_next(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&wsm[stage][p][(cbeg + c) * 32 + (lane & 31)][ks * 16 + 8 * h]);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][2], b[0], acc[c], 0, 0, 0);   // smallest terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][1], b[1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[2], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][1], b[0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[0], acc[c], 0, 0, 0);
            }
        if (kt + 1 < nk) {
            store_w(stage ^ 1);
            convert_x();
        }
        __syncthreads();
    }
}

struct EncF16Params {
    const float* x;
    const unsigned short* w2h;     // BlobHeader::enc_w2h
    const unsigned* w_bad;         // BlobHeader::enc_w2h_bad: kW2hBadWords words
    const unsigned short* w3;      // BlobHeader::enc_w3 (the bf16 arm's pieces)
    float* out;                    // split-K slabs [ks][M][128] (not FUSE)
    int M, K, kslice;
    int k_rotate;
    int force_arm;                 // diagnostics / tests: 1 = every tile takes the bf16 arm
    int prio_late_half;            // 1: waves 4-7 run the main loop at s_setprio 1
    int x_nt;                      // 1: the x stream's LDS-DMA requests carry the non-temporal cache policy
};

// DIAG (GNNCCA_DIAG builds of the ablation matrix, timing only -- the results are garbage): bit 0 = no MFMAs (the fragments stay used), bit 1 = every
// workgroup streams the FIRST 256 rows of x (x from L2), bit 2 = no fp16 split (the raw bits of x feed the MFMAs)
template <bool FUSE, int DIAG = 0>
__device__ __forceinline__ void enc_gemm_f16_body(const EncF16Params p, const EncFuseParams fp) {
    constexpr int BK = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int M = p.M, K = p.K;
    if (FUSE && blockIdx.x == gridDim.x - 1) {   // the plan workgroup (its first four waves: plan_finish is written for 256 threads)
        if (tid < 256)
            plan_finish(fp.ei, fp.E, M, fp.seg_ptr, fp.col32, fp.perm, fp.cursor, fp.flags, fp.blockflags, reinterpret_cast<unsigned*>(lds_raw));
        return;
    }
    PHASE_T_DECL;
    const int row0 = blockIdx.x * 256;
    const int kbeg = blockIdx.y * p.kslice;
    const int nk = min(p.kslice, K - kbeg) / BK;
    const int rot = p.k_rotate ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nk) : 0;
    auto kchunk = [&](int kt) {
        const int kr = kt + rot;
        return kr >= nk ? kr - nk : kr;
    };
    // ---- LDS-DMA sources: this workgroup's rows of x behind a descriptor of their own (rows beyond M read as zero: out of range),
    //      the whole fp16 W image behind another ---------------------------------------------------------------------------------------
    const int rows_here = min(256, M - row0);
    const rsrc_t rx = make_rsrc(p.x + ((DIAG & 2) ? (size_t)0 : (size_t)row0 * K), (unsigned long long)rows_here * K * 4);
    const rsrc_t rw = make_rsrc(p.w2h, (unsigned long long)(K / BK) * kF16WSlot);
    unsigned xoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = 8 * j + (lane >> 3);                           // row inside the wave's 32
        const int gs = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);      // source granule that lands in LDS granule slot (lane & 7) of that row
        xoff[j] = (unsigned)((size_t)(wave * 32 + rl) * K * 4 + gs * 16);
    }
    const unsigned woff = (unsigned)(wave * 2048 + lane * 16);
    unsigned char* xring = lds_raw + (size_t)wave * kF16Stages * kF16XSlot;
    unsigned char* wring = lds_raw + kF16XBytes;
    // six LDS-DMA instructions per wave and chunk: the wave's 32 rows x 128 B of x (j = 0 ... 3), and its eighth of the chunk's W image (4, 5)
    // (kt beyond the last chunk: a clamped duplicate into a stage nobody reads any more -- the loop stays branch-free and every iteration
    // leaves exactly six instructions in flight, which is what its one counted wait assumes)
    auto issue_one = [&](int kt, int j) {
        const int kc = kchunk(min(kt, nk - 1));
        const int st = kt % kF16Stages;
        if (j < 4) {
            if (p.x_nt)   // x is read once: the non-temporal policy keeps it from displacing what the step kernels left in the caches (and W)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_dma_ptr)(xring + st * kF16XSlot + j * 1024), 16, xoff[j & 3], (unsigned)(kbeg + kc * BK) * 4u, 0, 2);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_dma_ptr)(xring + st * kF16XSlot + j * 1024), 16, xoff[j & 3], (unsigned)(kbeg + kc * BK) * 4u, 0, 0);
        } else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_dma_ptr)(wring + st * kF16WSlot + wave * 2048 + (j - 4) * 1024), 16, woff + (j - 4) * 1024,
                                                     (unsigned)(kbeg / BK + kc) * (unsigned)kF16WSlot, 0, 0);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 6; ++j) issue_one(kt, j);
    };
    // weights beyond fp16: one word per lane of wave 0, folded with the x check below
    unsigned wbad = (wave == 0 && lane < kW2hBadWords) ? p.w_bad[lane] : 0u;
    issue(0);
    issue(1);
    f32x16 accA[4], accB[4];   // unscaled products x0 w0; products carrying one residual (x0 w1 + x1 w0), scaled by 2^11
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[c][i] = 0.f, accB[c][i] = 0.f;
    float amax = 0.f;
    // fragment addresses: A from the wave's fp32 ring (row l32, granules 4 s + 2 h and + 1, XOR (l32 >> 1) & 7); B from the W ring, whose
    // image is fragment-major (pack.cpp: w2h_index): piece p, column tile c, k-step s = the 1 KB at (p * 8 + c * 2 + s) * 1024, lane * 16 inside
    const int aswz = (l32 >> 1) & 7;
    const unsigned char* abase = xring + l32 * 128;
    // two waves share every SIMD; the later-dispatched half loses every arbitration at equal priority (stamps: its compute phase took
    // 2650 cycles per chunk against 1900, and the first half then waited for it at the barrier): one static priority for that half
    if (p.prio_late_half && wave >= 4) __builtin_amdgcn_s_setprio(1);
    PHASE_T(4);   // prologue
    for (int kt = 0; kt < nk; ++kt) {
        PHASE_T(3);
        // my own DMA of chunk kt has landed (issued two iterations ago; the six of chunk kt + 1 may still fly) ...
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        PHASE_T(0);
        // ... and so has every other wave's share of W(kt); every wave is also past its reads of stage (kt - 1) % 3, which the DMA below refills
        __builtin_amdgcn_s_barrier();
        PHASE_T(1);
        // The six DMA instructions of chunk kt + 2 are NOT issued here in one burst: the CU takes in an LDS-DMA instruction every ~36 cycles
        // (stamps: 48 of them after the barrier stalled waves 4-7 for 1700 cycles per chunk while their MFMAs waited behind them in program
        // order), so they go out one at a time between the MFMA groups below, where a full queue costs nothing
        PHASE_T(2);
        const int st = kt % kF16Stages;
        const unsigned char* a = abase + st * kF16XSlot;
        const unsigned char* b = wring + st * kF16WSlot;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(a + (((4 * s + 2 * h) ^ aswz) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(a + (((4 * s + 2 * h + 1) ^ aswz) << 4));
            f16x8 b0[4], b1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b0[c] = *reinterpret_cast<const f16x8*>(b + (c * 2 + s) * 1024 + lane * 16);
                b1[c] = *reinterpret_cast<const f16x8*>(b + 8192 + (c * 2 + s) * 1024 + lane * 16);
            }
            f16x8 a0, a1;
            if (DIAG & 4) {
                a0 = __builtin_bit_cast(f16x8, v0), a1 = __builtin_bit_cast(f16x8, v1);
            } else
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
            if (DIAG & 1) {
                asm volatile("" ::"v"(a0), "v"(a1), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
            } else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                accA[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0[c], accA[c], 0, 0, 0);
                accB[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1[c], accB[c], 0, 0, 0);
                accB[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0[c], accB[c], 0, 0, 0);
                const int slot = 4 * s + c;               // 0 ... 7: DMA j = slot - 1 behind the MFMAs of groups 1 ... 6
                if (slot >= 1 && slot <= 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_one(kt + 2, slot - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    PHASE_T(3);
    if (p.prio_late_half && wave >= 4) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped duplicates of the last two iterations must land before the rings are reused
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = fmaf(accB[c][i], 1.0f / 2048.0f, accA[c][i]);
    // ---- does this tile need the bf16 arm?  (workgroup-uniform: the arm has barriers) -------------------------------------------------
    __syncthreads();   // every wave is done with the rings
    unsigned* s_flag = reinterpret_cast<unsigned*>(lds_raw);
    if (tid == 0) *s_flag = 0u;
    __syncthreads();
    // bit 0: beyond fp16 (a NaN in x: the arm's business too); bit 1: somebody's |x| reaches 2^-10; bit 2: somebody's is not zero -- a tile
    // whose largest |x| is below 2^-10 without being zero takes the arm too (kF16Tiny, internal.h)
    {
        const unsigned f = ((!(amax < kF16Limit) || wbad != 0u || p.force_arm) ? 1u : 0u) | (amax >= kF16Tiny ? 2u : 0u) | (amax > 0.f ? 4u : 0u);
        if (f) atomicOr(s_flag, f);
    }
    __syncthreads();
    const unsigned fl_arm = *s_flag;
    const bool arm = (fl_arm & 1u) != 0u || (fl_arm & 6u) == 4u;
    __syncthreads();
    if (arm) enc_f16_bf16_arm(acc, p.x, p.w3, M, K, row0 + wave * 32, kbeg, nk, lds_raw);
    PHASE_T(5);   // drain, combine, arm decision
    if (!FUSE) {
        float* __restrict__ dst = p.out + (size_t)blockIdx.y * M * 128;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = c * 32 + l32;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M) dst[(size_t)row * 128 + col] = acc[c][i];
            }
        }
        PHASE_T(6);
        PHASE_T_FLUSH(7);
        return;
    }
    // ---- fused epilogue: h1 = [ReLU](acc + b1) -> LDS -> the rest of the encoder (enc_finish_tile_rows) -------------------------------
    constexpr int LD1 = 132;
    float* H1 = reinterpret_cast<float*>(lds_raw);   // [256][132] = 135 KB over the (dead) rings
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = c * 32 + l32;
        const float bias = fp.b1[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float v = acc[c][i] + bias;
            H1[rl * LD1 + col] = fp.relu_prev ? fmaxf(v, 0.f) : v;
        }
    }
    __syncthreads();
    enc_finish_tile_rows(H1, wave, l32, h, row0, M, fp);
    PHASE_T(6);   // epilogue
    PHASE_T_FLUSH(7);
}

// (concrete kernels around the body: with this toolchain a __global__ TEMPLATE that issues the LDS-DMA builtin from a lambda was not emitted)
__global__ __launch_bounds__(512) void enc_gemm_f16_fused_kernel(const EncF16Params p, const EncFuseParams fp) { enc_gemm_f16_body<true>(p, fp); }
__global__ __launch_bounds__(512) void enc_gemm_f16_split_kernel(const EncF16Params p, const EncFuseParams fp) { enc_gemm_f16_body<false>(p, fp); }
#ifdef GNNCCA_F16_ABLATIONS   // diagnostic twin build only (tools/ab_f16_ablations.sh)
#define GNNCCA_F16_DIAG_KERNEL(D) \
    __global__ __launch_bounds__(512) void enc_gemm_f16_fused_diag##D##_kernel(const EncF16Params p, const EncFuseParams fp) { enc_gemm_f16_body<true, D>(p, fp); }
GNNCCA_F16_DIAG_KERNEL(1)
GNNCCA_F16_DIAG_KERNEL(2)
GNNCCA_F16_DIAG_KERNEL(3)
GNNCCA_F16_DIAG_KERNEL(4)
GNNCCA_F16_DIAG_KERNEL(5)
GNNCCA_F16_DIAG_KERNEL(6)
GNNCCA_F16_DIAG_KERNEL(7)
#endif

}  // namespace gnncca

namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// The same layer on MID-SIZE batches (4096 ... ~24 000 nodes: the per-GPU share of BASELINE config 4 is 8192), round 5: 32 rows per workgroup,
// K split over the WAVES of the workgroup instead of over workgroups, the whole rest of the encoder in the epilogue.
// Up to round 4 these sizes ran the 256-row kernel split-K over the grid: eight [N][128] fp32 slabs to HBM and back (34 + 34 MB at N = 8192,
// next to 67 MB of x) and a tail launch (22-28 + 9 us).  Here a workgroup owns 32 rows and ALL of K: 8 compute waves = 4 k-quarters x 2 column
// halves, wave (kg, ch) accumulates x[32 rows][k quarter kg] . W[k quarter][64 columns of half ch]; the four k-quarters are summed through LDS
// in the epilogue, in fixed order; nothing goes to HBM but h0 and the step-1 projections, and no tail launch follows.
//   x: TWO LOADER WAVES (waves 8, 9; each serves two k-quarters) fill four rings of raw fp32 by LDS-DMA (32 rows x 128 B per 32-deep chunk
//      and quarter), five chunks ahead = 80 KB per CU in flight.  Loaders of their own because a wave's vector-memory operations retire IN
//      ORDER: a compute wave that waits for its next B operands (below) would thereby wait for every x request it issued before them, and
//      the x stream could never be more than one iteration deep.  One workgroup barrier per chunk publishes a chunk and frees a stage.
//   W: a compute wave's B operands are private (its k range, its 64 columns), so they never touch LDS: eight fully coalesced 1 KB loads per
//      chunk straight from the fragment-major fp16 image (BlobHeader::enc_w2h; L2-resident: every workgroup reads all 1 MB of it) into the
//      NEXT chunk's registers while the current chunk's MFMAs run.
//   arithmetic: the fp16-split form above (three products, two accumulators), the bf16 arm for out-of-range tiles.
// 256 workgroups at N = 8192: one per CU.
// ------------------------------------------------------------------------------------------------------------
#ifdef GNNCCA_STAMPS   // per-wave phase totals of up to ten waves: g_stamps[6][block < 1024][wave][16] (slot 6's region, 4 KB per block)
#define R32_STAMP_FLUSH                                                                                                             \
    do {                                                                                                                           \
        pt_acc[7] = __builtin_amdgcn_s_memrealtime() - pt_real0;                                                                   \
        if (g_stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                              \
            for (int q = 0; q < 8; ++q) g_stamps[(size_t)6 * 4096 * 4 * 16 + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + q] = pt_acc[q]; \
    } while (0)
#else
#define R32_STAMP_FLUSH do { } while (0)
#endif
constexpr int kF16R32Ahead = 5;                                       // chunks of x in flight per k-quarter
constexpr int kF16R32Stages = kF16R32Ahead + 1;
constexpr size_t kF16R32LdsBytes = 100 * 1024;                        // rings 4 x 6 x 4 KB = 96 KB; the epilogue overlays partials [4][32][128] f32 + H1 [32][132] f32
constexpr int kF16R32Threads = 640;                                   // 8 compute waves + 2 loader waves

__global__ __launch_bounds__(kF16R32Threads) void enc_gemm_f16_rows32_kernel(const EncF16Params p, const EncFuseParams fp) {
    constexpr int BK = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int M = p.M, K = p.K;
    if (blockIdx.x == gridDim.x - 1) {   // the plan workgroup
        if (tid < 256)
            plan_finish(fp.ei, fp.E, M, fp.seg_ptr, fp.col32, fp.perm, fp.cursor, fp.flags, fp.blockflags, reinterpret_cast<unsigned*>(lds_raw));
        return;
    }
    PHASE_T_DECL;
    const bool loader = wave >= 8;
    const int kg = (wave >> 1) & 3, ch = wave & 1;
    const int row0 = blockIdx.x * 32;
    const int nkc = K / BK / 4;                     // chunks per k-quarter
    const int rot = p.k_rotate ? (int)((blockIdx.x * 5u) % (unsigned)nkc) : 0;
    auto lchunk = [&](int it) {                    // the quarter-local chunk of iteration `it` (clamped past the end: a harmless duplicate)
        int lc = min(it, nkc - 1) + rot;
        return lc >= nkc ? lc - nkc : lc;
    };
    const int rows_here = min(32, M - row0);
    const rsrc_t rx = make_rsrc(p.x + (size_t)row0 * K, (unsigned long long)rows_here * K * 4);
    const rsrc_t rw = make_rsrc(p.w2h, (unsigned long long)(K / BK) * kF16WSlot);
    float amax = 0.f;
    unsigned wbad = (wave == 0 && lane < kW2hBadWords) ? p.w_bad[lane] : 0u;
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[c][i] = 0.f, accB[c][i] = 0.f;
    if (loader) {
        // ---- loader wave L serves k-quarters 2 L and 2 L + 1: eight 1 KB DMA instructions per chunk ------------------------------------
        const int L = wave - 8;
        unsigned xoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = 8 * j + (lane >> 3);
            const int gs = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);      // source granule that lands in LDS granule slot (lane & 7) of row rl
            xoff[j] = (unsigned)((size_t)rl * K * 4 + gs * 16);
        }
        auto issue = [&](int it) {
            const int st = it % kF16R32Stages, lc = lchunk(it);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int q = 2 * L + g;
                unsigned char* dst = lds_raw + ((size_t)q * kF16R32Stages + st) * 4096;
                const unsigned so = (unsigned)((q * nkc + lc) * BK) * 4u;
                if (p.x_nt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_dma_ptr)(dst + j * 1024), 16, xoff[j], so, 0, 2);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_dma_ptr)(dst + j * 1024), 16, xoff[j], so, 0, 0);
                }
            }
        };
#pragma unroll
        for (int d = 0; d < kF16R32Ahead; ++d) issue(d);
        PHASE_T(4);
        for (int it = 0; it < nkc; ++it) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // 8 x (kF16R32Ahead - 1): chunk `it` has landed, four later ones may fly
            PHASE_T(0);
            __builtin_amdgcn_s_barrier();                       // publishes x(it); every compute wave is past stage (it - 1) % 6 ...
            PHASE_T(1);
            issue(it + kF16R32Ahead);                              // ... which this refills
            PHASE_T(2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped duplicates must land before the rings are reused
        PHASE_T(5);
    } else {
        // ---- compute wave (kg, ch) --------------------------------------------------------------------------------------------------
        // B fragments of a chunk: [s][cc][piece], column tile c = 2 ch + cc.  ONE register set: a fragment is re-requested for the NEXT chunk
        // right behind the MFMAs that consumed it (hipcc counts the waits: the first fragments a chunk needs were requested a whole
        // iteration earlier, the last ones half of one) -- a second set would put this kernel over the 168 VGPRs that ten waves per
        // workgroup leave each of them, and any scratch slows a launch by half
        auto load_b1 = [&](int it, int s, int cc, int pc) {
            const unsigned base = (unsigned)(kg * nkc + lchunk(it)) * (unsigned)kF16WSlot;
            return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, base + (unsigned)(pc * 8192 + ((2 * ch + cc) * 2 + s) * 1024), 0));
        };
        f16x8 bq[2][2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) bq[s][cc][pc] = load_b1(0, s, cc, pc);
        const int aswz = (l32 >> 1) & 7;
        const unsigned char* abase = lds_raw + (size_t)kg * kF16R32Stages * 4096 + l32 * 128;
        PHASE_T(4);
        for (int it = 0; it < nkc; ++it) {
            PHASE_T(3);
            __builtin_amdgcn_s_barrier();                       // x(it) is published (the loader waited for it before arriving here)
            PHASE_T(1);
            const unsigned char* a = abase + (it % kF16R32Stages) * 4096;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(a + (((4 * s + 2 * h) ^ aswz) << 4));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(a + (((4 * s + 2 * h + 1) ^ aswz) << 4));
                f16x8 a0, a1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 v = q < 2 ? f32x2{v0[2 * q], v0[2 * q + 1]} : f32x2{v1[2 * q - 4], v1[2 * q - 3]};
                    const f16x2 p0 = __builtin_convertvector(v, f16x2);
                    const f32x2 r = (v - f32x2{(float)p0[0], (float)p0[1]}) * 2048.0f;
                    const f16x2 p1 = __builtin_convertvector(r, f16x2);
                    a0[2 * q] = p0[0], a0[2 * q + 1] = p0[1];
                    a1[2 * q] = p1[0], a1[2 * q + 1] = p1[1];
                    amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    accA[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[s][cc][0], accA[cc], 0, 0, 0);
                    accB[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[s][cc][1], accB[cc], 0, 0, 0);
                    accB[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq[s][cc][0], accB[cc], 0, 0, 0);
                    bq[s][cc][0] = load_b1(it + 1, s, cc, 0);
                    bq[s][cc][1] = load_b1(it + 1, s, cc, 1);
                }
            }
        }
        PHASE_T(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // the epilogue's operands from L2, requested NOW: their round trips run under the partial-tile exchange instead of after every barrier
    // (stamps: the epilogue took 8 800 cycles per wave, most of it three dependent L2 latencies)
    float w2v[16], pjv[16];
    float bias2_pre = 0.f, pb_pre = 0.f;
    auto request_epilogue_operands = [&](const float* __restrict__ W2, const float* __restrict__ projwT) {
        const int q = wave & 3;
        const float* w2row = W2 + (size_t)l32 * 128 + 32 * q + 16 * h;
        const int slot = 32 * (wave & 1) + l32;
        const bool on = slot < kProjOut;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = wave < 4 ? *reinterpret_cast<const f32x4*>(w2row + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) w2v[4 * j + t] = b4[t];
        }
#pragma unroll
        for (int sI = 0; sI < 16; ++sI) pjv[sI] = (wave < 2 && on) ? projwT[(16 * h + sI) * kProjOut + slot] : 0.f;
        if (wave == 0) bias2_pre = fp.b2[l32];
        if (wave < 2 && on) pb_pre = fp.projb[slot];
    };
    request_epilogue_operands(fp.W2, fp.projwT);
    // ---- the four k-quarters' partial tiles go to LDS at once (the accumulators die here: the bf16 arm below needs the registers), then:
    //      bf16 arm?  (workgroup-uniform) ----------------------------------------------------------------------------------------------
    __syncthreads();                                              // every wave is done with the rings
    constexpr int LD1 = 132;
    float* P = reinterpret_cast<float*>(lds_raw);                 // [4 k-quarters][32][128] partial tiles
    float* H1 = P + 4 * 32 * 128;                                 // [32][132]
    unsigned* s_flag = reinterpret_cast<unsigned*>(lds_raw + kF16R32LdsBytes - 16);
    if (tid == 0) *s_flag = 0u;
    if (!loader) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = (2 * ch + cc) * 32 + l32;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
                P[(kg * 32 + rl) * 128 + col] = fmaf(accB[cc][i], 1.0f / 2048.0f, accA[cc][i]);
            }
        }
    }
    __syncthreads();
    {   // (bits as in the 256-row kernel: 0 beyond fp16, 1 somebody's |x| reaches 2^-10, 2 somebody's is not zero)
        const unsigned f = ((!(amax < kF16Limit) || wbad != 0u || p.force_arm) ? 1u : 0u) | (amax >= kF16Tiny ? 2u : 0u) | (amax > 0.f ? 4u : 0u);
        if (f) atomicOr(s_flag, f);
    }
    __syncthreads();
    const unsigned fl_arm = *s_flag;
    const bool arm = (fl_arm & 1u) != 0u || (fl_arm & 6u) == 4u;
    if (arm) {
        __syncthreads();                                          // (the arm's W stages overwrite the partials)
        // every compute wave recomputes the SAME 32 x 128 tile over all of K on the bf16 arm (its W staging wants all 512 compute threads;
        // rare by construction); wave 0's copy is the tile.  (The arm's stages end below H1.)
        for (int cb = 0; cb < 4; cb += 2) {                       // (two column halves: see enc_f16_bf16_arm on NC)
            f32x16 part[2];
            enc_f16_bf16_arm<2>(part, p.x, p.w3, M, K, row0, 0, K / BK, lds_raw, !loader, cb);
            if (wave == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int col = (cb + c) * 32 + l32;
                    const float bias = fp.b1[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
                        const float v = part[c][i] + bias;
                        H1[rl * LD1 + col] = fp.relu_prev ? fmaxf(v, 0.f) : v;
                    }
                }
            }
        }
    } else {
        // h1 = [ReLU](((p0 + p1) + p2) + p3 + b1): 1024 float4, two per compute thread
        if (!loader) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e4 = tid + 512 * u;                      // float4 index in [32][128 / 4]
                const int rl = e4 >> 5, c4 = (e4 & 31) * 4;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(P + (0 * 32 + rl) * 128 + c4);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(P + (1 * 32 + rl) * 128 + c4);
                const f32x4 q2 = *reinterpret_cast<const f32x4*>(P + (2 * 32 + rl) * 128 + c4);
                const f32x4 q3 = *reinterpret_cast<const f32x4*>(P + (3 * 32 + rl) * 128 + c4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(fp.b1 + c4);
                f32x4 v = ((q0 + q1) + q2) + q3 + b;
                if (fp.relu_prev) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
                }
                *reinterpret_cast<f32x4*>(H1 + rl * LD1 + c4) = v;
            }
        }
    }
    __syncthreads();
    // ---- layer 2 and the projections on the 32-row tile: the arithmetic of enc_finish_tile_rows with its four k-quarters dealt to waves 0-3
    //      (partial tiles summed ((d0 + d1) + d2) + d3 by wave 0, as every other encoder kernel does) and the two projection slot groups to
    //      waves 0-1 ----------------------------------------------------------------------------------------------------------------------
    float* DQ = P;                                                // [4][16 regs][64 lanes]: the partial tiles in accumulator layout
    if (wave < 4) {
        const int q = wave;
        const float* hrow = H1 + l32 * LD1 + 32 * q + 16 * h;
        float av[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hrow + 4 * j);
#pragma unroll
            for (int t = 0; t < 4; ++t) av[4 * j + t] = a4[t];
        }
        f32x16 dq;
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], w2v[s], dq, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) DQ[(q * 16 + i) * 64 + lane] = dq[i];
    }
    __syncthreads();
    constexpr int LD0 = 36;
    float* H0 = H1;                                               // [32][36] over the consumed h1 tile
    if (wave == 0) {
        const float bias2 = bias2_pre;
        float hv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = ((DQ[(0 * 16 + i) * 64 + lane] + DQ[(1 * 16 + i) * 64 + lane]) + DQ[(2 * 16 + i) * 64 + lane]) + DQ[(3 * 16 + i) * 64 + lane];
            hv[i] = fmaxf(d + bias2, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
            H0[rl * LD0 + l32] = hv[i];
            const int row = row0 + rl;
            if (row < M) {
                fp.h0[(size_t)row * kH + l32] = hv[i];
                if (fp.trace_h) fp.trace_h[(size_t)row * kH + l32] = hv[i];
            }
        }
    }
    __syncthreads();
    if (wave < 2) {
        const int t = wave;
        float a2[16];
        const float* hr = H0 + l32 * LD0 + 16 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) a2[4 * j + q] = a4[q];
        }
        const int slot = 32 * t + l32;
        const bool on = slot < kProjOut;
        f32x16 pacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], pjv[s], pacc, 0, 0, 0);
        const float pb = pb_pre;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (row < M && on) {
                const float v = pacc[i] + pb;
                if (slot < kPdStride)
                    fp.pd_out[(size_t)row * kPdStride + slot] = v;
                else
                    fp.psq_out[(size_t)row * kPsQStride + slot - kPdStride] = v;
            }
        }
    }
    PHASE_T(6);
    R32_STAMP_FLUSH;
}

}  // namespace gnncca
