#pragma once
// Part of the single translation unit mpn_forward.hip (kernels share device helpers and the launch code below
// instantiates their templates); see that file for the overall picture.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// Node encoder GEMM: part[ks][M][O] = in[M][kslice ks] . W[O][kslice ks]^T with v_mfma_f32_32x32x2_f32 (exact
// fp32 FMA chain).  One wave = 32 rows x 32 output columns; a workgroup = 4 waves = 128 columns.
// k-permutation: within a 64-deep chunk, lane (r, h) feeds k = kc + 32h + s at MFMA step s for BOTH operands, so
// every lane reads 128 contiguous bytes of its own row (8 x float4) and no LDS transpose is needed.
// Split-K fills the chip when M is small (M = 256 nodes -> 8 row tiles x 32 slices).
// Replaces the first nn.Linear of encoder.node_mlp (models/mpn.py:131 <- models/mlp.py:13).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gemm_tile(int rt, int ks, int cg, const float* __restrict__ in, const float* __restrict__ W,
                                          float* __restrict__ part, int M, int K, int O, int kslice, int vec_ok) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = rt * 32;
    const int col0 = (cg * 4 + wave) * 32;
    if (col0 >= O) return;
    const int arow = min(row0 + r, M - 1);
    const int wrow = min(col0 + r, O - 1);
    const float* __restrict__ ap = in + (size_t)arow * K;
    const float* __restrict__ wp = W + (size_t)wrow * K;
    const int kbeg = ks * kslice;
    const int kend = min(kbeg + kslice, K);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int kc = kbeg; kc < kend; kc += 64) {
        float a[32], b[32];
        const int k0 = kc + 32 * h;
        if (vec_ok && kc + 64 <= kend) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + k0 + 4 * j);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wp + k0 + 4 * j);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[4 * j + q] = av[q];
                    b[4 * j + q] = bv[q];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int k = k0 + s;
                a[s] = (k < kend) ? ap[k] : 0.f;
                b[s] = (k < kend) ? wp[k] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    const int col = col0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M && col < O) part[((size_t)ks * M + row) * O + col] = acc[i];
    }
}

// Second half of the plan, run by ONE 256-thread workgroup of the launch that follows the plan blocks on the stream:
// One launch, two roles: workgroups [0, gemm_blocks) run encoder GEMM tiles, the rest run the graph plan -- the
// two are independent, so the plan's HBM pass over edge_index hides under the GEMM instead of costing a launch.
struct EncPlanParams {
    const float* in;
    const float* W;
    float* part;
    const long long* ei;
    int* seg_ptr;
    int* col32;
    unsigned* blockflags;
    int M, K, O, kslice, vec_ok, nrt, nks, gemm_blocks, E, N;
    int ell_S;  // slots per node of the padded step layout to validate the degrees against (0: none)
    int plan_span;  // 1024-edge plan blocks per plan workgroup (plan.cuh: 1, or kPlanSpan on a plan-only launch of a big batch)
};

__global__ __launch_bounds__(256) void enc_gemm_plan_kernel(const EncPlanParams p) {
    __shared__ unsigned s_fl;
    GNNCCA_STAMP(1, 0);
    const int b = blockIdx.x;
    if (b < p.gemm_blocks) {
        const int rt = b % p.nrt, t = b / p.nrt;
        gemm_tile(rt, t % p.nks, t / p.nks, p.in, p.W, p.part, p.M, p.K, p.O, p.kslice, p.vec_ok);
    } else {
        plan_block(b - p.gemm_blocks, p.ei, p.E, p.N, p.seg_ptr, p.col32, p.blockflags, &s_fl, p.ell_S, p.plan_span);
    }
    GNNCCA_STAMP(1, 1);
}

// The plan alone (big batches give it a launch of its own): the same plan_block, without the GEMM tile's registers in the kernel's
// budget -- enc_gemm_plan_kernel allocates 92 VGPRs (five waves per SIMD), a pure stream wants every wave slot it can get.
__global__ __launch_bounds__(256) void plan_only_kernel(const EncPlanParams p) {
    __shared__ unsigned s_fl;
    plan_block(blockIdx.x, p.ei, p.E, p.N, p.seg_ptr, p.col32, p.blockflags, &s_fl, p.ell_S, p.plan_span);
}

// ------------------------------------------------------------------------------------------------------------
// Large-N encoder GEMM on the bf16 MFMA pipe with fp32-level accuracy ("split-bf16"):  x = x0 + x1 + x2 and
// w = w0 + w1 + w2 with bf16 pieces (3 x 8 = 24 mantissa bits, the pieces of w prepared at pack time, those of x
// on the fly), and  x.w ~= x2w0 + x1w1 + x0w2 + x1w0 + x0w1 + x0w0  -- the dropped terms are <= 2^-24 relative.
// Every product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32: measured against fp64 the
// result is MORE accurate than an fp32 GEMM (1.9e-9 vs 1.4e-7 on the N=64 golden case), while the six
// v_mfma_f32_32x32x16_bf16 cost 6/16 of the v_mfma_f32_32x32x2_f32 time.  (Keeping only x0w0 + x0w1 + x1w0 would
// halve the MFMA work again -- GNNCCA_OPT_ENC_SPLIT3, off by default.  Measured (tools/archive/exp_enc_products.py, x ~ N(0,1),
// N = 8 192 / 51 233): encoder output 5.0e-6 / 5.5e-6 from an fp64 evaluation instead of 3.0e-7 / 9.9e-7 (an fp32 GEMM:
// 4.9e-7 / 5.7e-7), logits 0.9e-7 / 1.5e-7 from the fp32 oracle instead of 0.6e-7 (tolerance 1e-4); GEMM 30.8 -> 22.3 us at
// N = 8 192, 57.7 -> 44.7 at 16 384, 225 -> 183 at 65 536.)
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// Structure ("direct A"): a wave owns 32 rows x all 128 columns, so its A operand
// never needs sharing -- x goes global -> registers -> three bf16 fragments (converted once per element) and
// never touches LDS.  Only the weight tile goes through LDS, double-buffered, ONE barrier per 32-deep chunk:
//   top of iteration : global loads of the next chunk's W pieces (6 x 16 B per lane) and x (4 x 16 B per lane)
//   middle           : 2 k-steps x 4 column tiles x 6 MFMAs from the current LDS stage + the A fragments in registers
//   bottom           : next W -> the other LDS stage, next x -> A fragments (loads had the whole middle to land)
// Workgroup = 4 waves = 128 rows.  Split-K over blockIdx.y for mid-size batches.
// ------------------------------------------------------------------------------------------------------------
// Grid: 1-D.  Workgroup b < rb * ks is GEMM tile (row block b % rb, k slice b / rb); the workgroups beyond run the graph plan
// (plan.cuh) when `plan.E > 0`: on graphs of 1024...4095 nodes the plan rides in this launch as it does in enc_gemm_plan_kernel's
// below 1024 (a launch of its own cost 4.5-17 us there).
template <bool P3>
__global__ __launch_bounds__(256) void enc_gemm_split_direct_kernel(const float* __restrict__ x,
                                                                    const unsigned short* __restrict__ w3,
                                                                    float* __restrict__ out, int M, int K, int O, int kslice, int rb, int ks_n,
                                                                    const EncPlanParams plan) {
    constexpr int BN = 128, BK = 32, LDK = 40;
    __shared__ __attribute__((aligned(16))) __bf16 wsm[2][3][BN][LDK];
    __shared__ unsigned s_plan_fl;
    if ((int)blockIdx.x >= rb * ks_n) {
        plan_block((int)blockIdx.x - rb * ks_n, plan.ei, plan.E, plan.N, plan.seg_ptr, plan.col32, plan.blockflags, &s_plan_fl, plan.ell_S,
                   plan.plan_span);
        return;
    }
    const int bx = (int)blockIdx.x % rb, by = (int)blockIdx.x / rb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = bx * 128 + wave * 32;
    const int kbeg = by * kslice;
    const size_t plane = (size_t)O * BK;  // w3 is [K/32][3 pieces][O][32]: a chunk's 24 KB are contiguous
    const int h = lane >> 5;
    const float* __restrict__ xrow = x + (size_t)min(row0 + (lane & 31), M - 1) * K + kbeg + 8 * h;
    int wp[6], wcol[6], wk[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int idx = tid + 256 * u;
        wp[u] = idx >> 9;
        wcol[u] = (idx & 511) >> 2;
        wk[u] = (idx & 3) * 8;
    }
    f32x4 xreg[4];   // chunk's x: k-step 0 -> [0],[1]; k-step 1 -> [2],[3]   (8 consecutive k each)
    bf16x8 wreg[6];
    bf16x8 afrag[2][3];
    const int nk = min(kslice, K - kbeg) / BK;
    auto load_next = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xreg[2 * ks] = *reinterpret_cast<const f32x4*>(xrow + kt * BK + ks * 16);
            xreg[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xrow + kt * BK + ks * 16 + 4);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u)
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
                afrag[ks][0][q] = h0;
                afrag[ks][1][q] = h1;
                afrag[ks][2][q] = (__bf16)r2;
            }
    };
    auto store_w = [&](int stage) {
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<bf16x8*>(&wsm[stage][wp[u]][wcol[u]][wk[u]]) = wreg[u];
    };
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    load_next(0);
    store_w(0);
    convert_x();
    __syncthreads();
    const int k8 = 8 * h;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) load_next(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[p] = *reinterpret_cast<const bf16x8*>(&wsm[stage][p][c * 32 + (lane & 31)][ks * 16 + k8]);
                // smallest terms first
                if (!P3) {   // the three 2^-16-order terms (GNNCCA_OPT_ENC_SPLIT3 drops them)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][2], b[0], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][1], b[1], acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[2], acc[c], 0, 0, 0);
                }
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][1], b[0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks][0], b[0], acc[c], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            store_w(stage ^ 1);
            convert_x();
        }
        __syncthreads();
    }
    float* __restrict__ dst = out + (size_t)by * M * O;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = c * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (row < M) dst[(size_t)row * O + col] = acc[c][i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Structure ("both operands through LDS", N >= 16 K rows): rocprofv3 counters on the direct-A kernel above showed the
// L1 (TCP) as its limit -- 93 % active, TA 68 % busy and 35 % stalled by it: a fragment-shaped x load (lane = row)
// touches 64 cache lines for 1 KB, eight times the line accesses of a full-line load.  Here x is read in full 128-B
// lines (a wave instruction = 8 rows x 128 B), converted ONCE to its three bf16 pieces by the loading thread and
// written to LDS in the fragment image; W arrives the same way from its chunk-contiguous blob layout.
//   workgroup = 8 waves = 256 rows x 128 columns, wave = 64 x 64 (2 x 2 MFMA tiles: A and B fragments each reused
//   twice), BK = 32, two LDS stages of (A 48 KB + B 24 KB), one barrier per chunk.
//   LDS image: [stage][piece][row][32 k] bf16, 64-B rows, the 16-B granule kc of row r stored at kc ^ ((r >> 2) & 3):
//   ds_read_b128 of 16 consecutive rows at one kc hits 64 distinct banks (no padding: 144 KB must fit 160 KB).
// Measured (N = 65 536, counters): TCP line accesses 52.9 M -> 18.4 M, TA busy 82 M -> 32 M cycles, LDS bank
// conflicts 0.  Time, A/B in one process on one box (GNNCCA_GEMM_DIRECT=1 selects the kernel above): 195-197 us ->
// 189-192 us at N = 65 536, 59.9 -> 57.4 us at N = 16 384 -- the L1 pressure was real but is not what bounds the
// kernel.  Requesting the operands two chunks ahead (two register sets, 215 VGPRs) did not help either kernel (+4 %
// here): what remains is not load latency either; the waves spend 60 % of their cycles waiting on instruction issue
// with the MFMA pipe ~50 % busy, i.e. the dependent six-product MFMA chains and the per-chunk barrier.
// ------------------------------------------------------------------------------------------------------------
constexpr int kLdsGemmRows = 256;
constexpr size_t kLdsGemmBytes = (size_t)2 * 3 * (kLdsGemmRows + 128) * 32 * 2;  // 147 456

// What the fused epilogue of the un-split launch needs (enc_gemm_split_lds_kernel<true>): the rest of encoder.node_mlp and the
// step-1 projections, applied to the 256 x 128 tile while it is still on chip -- no [N][128] partial goes to HBM and comes
// back, and the wave-per-node tail kernel (63 us at N = 65 536, 0.11 of HBM peak) disappears from that regime.
struct EncFuseParams {
    const float* b1;       // [128]      bias of the first encoder layer
    const float* W2;       // [32][128]  last encoder layer, row-major [out][in]
    const float* b2;       // [32]
    const float* projwT;   // [32][48]   per-node projection, k-major (BlobHeader::proj_wT)
    const float* projb;    // [48]
    float* h0;             // [N][32]
    float* trace_h;        // [N][32] or null
    float* pd_out;         // [N][8]
    float* psq_out;        // [N][40]
    int relu_prev;
    // second half of the graph plan (plan_finish), run by the LAST workgroup of the launch -- the plan blocks were launched before
    const long long* ei;
    int* seg_ptr;
    int* col32;
    int* perm;
    int* cursor;
    unsigned* flags;
    const unsigned* blockflags;
    int E;
    int k_rotate;      // 1: workgroup b walks its k chunks from chunk (37 b) mod nk on, wrapping (the sum is order-free; see the kernel)
    int diag_x_rows;   // GNNCCA_DIAG experiment (0 in production): > 0 = every workgroup streams rows [0, diag_x_rows) of x instead of its
                       // own -- the same loads, conversions and MFMAs with x served from L2 (timing only: the results are garbage)
};

// Main loop: top of iteration kt = global loads of chunk kt + 1 (x: 4 x 16 B per thread, W pieces: 3 x 16 B), middle = the
// MFMAs of chunk kt from LDS stage kt & 1, bottom = conversion + LDS store of chunk kt + 1, one barrier.  Measured and NOT
// kept (round 2, N = 65 536, same box): waves 4-7 running [convert][MFMA] against waves 0-3 [MFMA][convert] (the stagger of
// MI355X_MICROARCH's "two waves that run the same program") 202 -> 220 us; x two chunks ahead in a second register set
// 202 -> 216 us.  Ablations of the same build: no MFMAs and no LDS reads at all 160 us; x re-read from L2 instead of HBM 183 us;
// no conversion / LDS stores 183 us -- the kernel is bound by its x stream, not by the matrix pipe: walking a 256-row tile in
// 128-B-per-row chunks with one workgroup per CU streams at 4.4 TB/s even in a bare copy loop (tools/ubench_rowtile.hip:
// 122 us for these 537 MB; a linear sweep of the same bytes 86 us).
// PIPE (round 2): the same operands, stages and MFMAs, software-pipelined by k-HALF with the chunk's one barrier BETWEEN the halves.
// With the barrier at the chunk end (the form above) every wave of the workgroup converts / stores / reads fragments at the same
// time and then queues for the matrix pipe at the same time -- 48 % MFMA-busy, and an LDS-DMA variant that hid ALL of the x latency
// ran no faster.  Here, while the 24 MFMAs of one half run, the wave reads the fragments of the next half
// (the next chunk's first half included: that chunk is published at the mid-chunk barrier) and converts + stores the chunk after
// it; x is requested two chunks ahead in two register sets.
template <bool FUSE, bool P3, bool PIPE = false>
__global__ __launch_bounds__(512) void enc_gemm_split_lds_kernel(const float* __restrict__ x, const unsigned short* __restrict__ w3,
                                                                 float* __restrict__ out, int M, int K, int O, int kslice,
                                                                 const EncFuseParams fp) {
    constexpr int BK = 32, RA = kLdsGemmRows, RB = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* sa = reinterpret_cast<__bf16*>(lds_raw);                 // [2][3][RA][32]
    __bf16* sb = sa + (size_t)2 * 3 * RA * BK;                       // [2][3][RB][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, l32 = lane & 31;
    if (FUSE && blockIdx.x == gridDim.x - 1) {   // the plan workgroup (its first four waves: plan_finish is written for 256 threads)
        if (tid < 256)
            plan_finish(fp.ei, fp.E, M, fp.seg_ptr, fp.col32, fp.perm, fp.cursor, fp.flags, fp.blockflags,
                        reinterpret_cast<unsigned*>(lds_raw));
        return;
    }
    const int row0 = blockIdx.x * RA;
    const int kbeg = blockIdx.y * kslice;
    const int nk = min(kslice, K - kbeg) / BK;
    // k ROTATION (round 5; an option, off by default: mpn_forward.hip says why).  Every workgroup of a round reads 128 B from each of its 256 rows (8 KB apart) per chunk, and all of them walk
    // the same k at about the same time: the chip's 256 request streams then agree in the address bits BELOW the row stride chunk after
    // chunk, i.e. they crowd the same memory channels (tools/ubench_xring.hip: this walk streams at 5.0 TB/s, the same walk with every
    // workgroup STARTING at another chunk at 6.2 -- the linear-sweep rate).  A dot product does not care where its sum starts, so
    // workgroup b takes its chunks in the order rot, rot + 1, ..., nk - 1, 0, ..., rot - 1 with rot = 37 b mod nk.  The summation ORDER
    // then depends on the row block, which is why GNNCCA_OPT_ENC_UNSPLIT (bitwise batch independence) runs with k_rotate = 0.
    const int rot = fp.k_rotate ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nk) : 0;
    auto kchunk = [&](int kt) {
        const int kr = kt + rot;
        return kr >= nk ? kr - nk : kr;
    };
    // loader roles: x chunk = 256 rows x 8 float4 -> 4 per thread; W chunk = 1536 16-B granules -> 3 per thread
    const float* xsrc[4];
    int xdst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 512 * u, row = q >> 3, c4 = q & 7;
        const int xrow = fp.diag_x_rows > 0 ? row % fp.diag_x_rows : min(row0 + row, M - 1);
        xsrc[u] = x + (size_t)xrow * K + kbeg + c4 * 4;
        xdst[u] = row * BK + (((c4 >> 1) ^ ((row >> 2) & 3)) << 3) + ((c4 & 1) << 2);
    }
    int wdst[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int g = tid + 512 * u, p = g >> 9, col = (g & 511) >> 2, kc = g & 3;
        wdst[u] = (p * RB + col) * BK + ((kc ^ ((col >> 2) & 3)) << 3);
    }
    const size_t wchunk = (size_t)3 * O * BK;  // bf16 elements per k-chunk of w3
    f32x4 xr[4];
    bf16x8 wreg[3];
    auto load_x = [&](int kt) {
        const int kc = kchunk(kt);
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u] = *reinterpret_cast<const f32x4*>(xsrc[u] + kc * BK);
    };
    auto load_w = [&](int kt) {
        const int kc = kchunk(kt);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            wreg[u] = *reinterpret_cast<const bf16x8*>(w3 + (size_t)(kbeg / BK + kc) * wchunk + (size_t)(tid + 512 * u) * 8);
    };
    auto store_stage = [&](int stage) {
        __bf16* a = sa + (size_t)stage * 3 * RA * BK;
        __bf16* b = sb + (size_t)stage * 3 * RB * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
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
            *reinterpret_cast<bf16x4*>(a + xdst[u]) = p0;
            *reinterpret_cast<bf16x4*>(a + RA * BK + xdst[u]) = p1;
            *reinterpret_cast<bf16x4*>(a + 2 * RA * BK + xdst[u]) = p2;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<bf16x8*>(b + wdst[u]) = wreg[u];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
    // fragment addresses (elements), per row tile / column tile; the k granule is kc = 2 ks + h
    int arow[2], brow[2], aswz[2], bswz[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wr * 64 + t * 32 + l32, cb = wc * 64 + t * 32 + l32;
        arow[t] = ra * BK, aswz[t] = (ra >> 2) & 3;
        brow[t] = cb * BK, bswz[t] = (cb >> 2) & 3;
    }
    auto mfma_chunk = [&](int stage) {
        const __bf16* a = sa + (size_t)stage * 3 * RA * BK;
        const __bf16* b = sb + (size_t)stage * 3 * RB * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = 2 * ks + h;
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[t][p] = *reinterpret_cast<const bf16x8*>(a + p * RA * BK + arow[t] + ((kc ^ aswz[t]) << 3));
                    bf[t][p] = *reinterpret_cast<const bf16x8*>(b + p * RB * BK + brow[t] + ((kc ^ bswz[t]) << 3));
                }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    // smallest terms first
                    if (!P3) {   // the three 2^-16-order terms (GNNCCA_OPT_ENC_SPLIT3 drops them)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][2], bf[ct][0], acc[rt][ct], 0, 0, 0);
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[ct][1], acc[rt][ct], 0, 0, 0);
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ct][2], acc[rt][ct], 0, 0, 0);
                    }
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[ct][0], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ct][1], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[ct][0], acc[rt][ct], 0, 0, 0);
                }
        }
    };
    if (!PIPE) {
        load_w(0);
        load_x(0);
        store_stage(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                load_w(kt + 1);
                load_x(kt + 1);
            }
            mfma_chunk(kt & 1);
            if (kt + 1 < nk) store_stage((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        struct Frag {
            bf16x8 a[2][3], b[2][3];
        };
        auto read_frag = [&](int stage, int ks, Frag& f) {
            const __bf16* a = sa + (size_t)stage * 3 * RA * BK;
            const __bf16* b = sb + (size_t)stage * 3 * RB * BK;
            const int kc = 2 * ks + h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    f.a[t][p] = *reinterpret_cast<const bf16x8*>(a + p * RA * BK + arow[t] + ((kc ^ aswz[t]) << 3));
                    f.b[t][p] = *reinterpret_cast<const bf16x8*>(b + p * RB * BK + brow[t] + ((kc ^ bswz[t]) << 3));
                }
        };
        auto mfma_half = [&](const Frag& f) {
#ifdef GNNCCA_EXP_GEMM_NO_MFMA   // diagnostic twin build: the main loop without its MFMAs (the fragments stay used: their LDS reads remain)
            asm volatile("" ::"v"(f.a[0][0]), "v"(f.a[1][2]), "v"(f.b[0][0]), "v"(f.b[1][2]));
            return;
#endif
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if (!P3) {
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][2], f.b[ct][0], acc[rt][ct], 0, 0, 0);
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][1], f.b[ct][1], acc[rt][ct], 0, 0, 0);
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][0], f.b[ct][2], acc[rt][ct], 0, 0, 0);
                    }
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][1], f.b[ct][0], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][0], f.b[ct][1], acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rt][0], f.b[ct][0], acc[rt][ct], 0, 0, 0);
                }
        };
        // x in two register sets (chunk k in set k & 1), W in one (its source is L2-resident)
        f32x4 xs[1][4];
        auto load_x2 = [&](int kt, f32x4 (&dst)[4]) {
            const int kc = kchunk(kt);
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[u] = *reinterpret_cast<const f32x4*>(xsrc[u] + kc * BK);
        };
        auto store_stage2 = [&](int stage, const f32x4 (&src)[4], const bf16x8 (&wsrc)[3]) {
            __bf16* a = sa + (size_t)stage * 3 * RA * BK;
            __bf16* b = sb + (size_t)stage * 3 * RB * BK;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // the same three roundings as store_stage, two elements at a time: v_cvt_pk_bf16_f32 rounds a pair, the pair's
                // float images are one shift and one mask of that word, the remainders one packed subtract (exact: a value minus
                // its own 8-bit rounding fits fp32) -- 9 VALU per pair instead of 13
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                unsigned w[3][2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x2 v = {src[u][2 * pr], src[u][2 * pr + 1]};
#pragma unroll
                    for (int lev = 0; lev < 3; ++lev) {
                        const unsigned word = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                        w[lev][pr] = word;
                        if (lev < 2)
                            v = __builtin_elementwise_fma(f32x2{-1.f, -1.f},
                                                          f32x2{__uint_as_float(word << 16), __uint_as_float(word & 0xffff0000u)}, v);
                    }
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(a + xdst[u]) = u32x2{w[0][0], w[0][1]};
                *reinterpret_cast<u32x2*>(a + RA * BK + xdst[u]) = u32x2{w[1][0], w[1][1]};
                *reinterpret_cast<u32x2*>(a + 2 * RA * BK + xdst[u]) = u32x2{w[2][0], w[2][1]};
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) *reinterpret_cast<bf16x8*>(b + wdst[u]) = wsrc[u];
        };
        Frag f0, f1;
        // chunk k lives in stage k & 1.  Phase 1 of chunk kt: MFMAs of its first half while its second half is read.  Barrier: every
        // wave has read all of chunk kt (its stage is free) and chunk kt + 1 -- stored a chunk ago -- is visible.  Phase 2: MFMAs of
        // the second half while chunk kt + 2 is converted and stored into the freed stage, the loads of chunk kt + 3 are issued (they
        // have a whole chunk to land: an HBM round trip under load is about one) and the first half of chunk kt + 1 is read.
        // (Requesting chunks 0 and 1 together, chunk 1 in a register set of its own, changes nothing: 28.3 vs 27.7-29 us at N = 8192,
        // r3_prol1.log -- the launch's 8.5 us of fixed cost are its 128 KB of slab stores per workgroup, not the prologue's round trips.)
        load_w(0);
        load_x2(0, xs[0]);
        store_stage2(0, xs[0], wreg);
        load_w(min(1, nk - 1));
        load_x2(min(1, nk - 1), xs[0]);
        store_stage2(1, xs[0], wreg);
        load_w(min(2, nk - 1));
        load_x2(min(2, nk - 1), xs[0]);
        __syncthreads();
        read_frag(0, 0, f0);
        PHASE_T_DECL;
        for (int kt = 0; kt < nk - 1; ++kt) {   // every chunk but the last: straight-line phases (loads unconditional, clamped)
            const int stage = kt & 1;
            PHASE_T(3);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(stage, 1, f1);
            mfma_half(f0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, P3 ? 1 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            PHASE_T(0);
            __syncthreads();
            PHASE_T(1);
            store_stage2(stage, xs[0], wreg);            // chunk kt + 2 (for kt + 2 >= nk: a clamped duplicate nobody reads)
            load_w(min(kt + 3, nk - 1));
            load_x2(min(kt + 3, nk - 1), xs[0]);
            read_frag(stage ^ 1, 0, f0);
            mfma_half(f1);
#pragma unroll
            for (int i = 0; i < (P3 ? 12 : 24); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, P3 ? 10 : 5, 1);   // conversion arithmetic
                __builtin_amdgcn_sched_group_barrier(0x080, P3 ? 3 : 2, 1);    // LDS stores / fragment reads
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);   // a global load
            }
            __builtin_amdgcn_sched_barrier(0);
            PHASE_T(2);
        }
        PHASE_T_FLUSH(6);
        {   // last chunk: nothing left to stage
            read_frag((nk - 1) & 1, 1, f1);
            mfma_half(f0);
            mfma_half(f1);
        }
        __syncthreads();   // the epilogue overwrites the stages
    }
    if (!FUSE) {
        float* __restrict__ dst = out + (size_t)blockIdx.y * M * O;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = wc * 64 + ct * 32 + l32;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0 + wr * 64 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (row < M) dst[(size_t)row * O + col] = acc[rt][ct][i];
                }
            }
        return;
    }
    // ---- fused epilogue (O == 128, un-split): h1 = [ReLU](acc + b1) -> LDS -> h0 = ReLU(h1 W2^T + b2) -> projections ----------
    // every MFMA below is v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, as in the tail kernels
    constexpr int LD1 = 132, LD0 = 36;   // row strides (floats): ds_read_b128 of 16 consecutive rows hits 64 distinct banks
    float* H1 = reinterpret_cast<float*>(lds_raw);   // [256][132] = 135 KB, over the (dead) stages: the loop ended with a barrier
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = wc * 64 + ct * 32 + l32;
            const float bias = fp.b1[col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = wr * 64 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = acc[rt][ct][i] + bias;
                H1[rl * LD1 + col] = fp.relu_prev ? fmaxf(v, 0.f) : v;
            }
        }
    __syncthreads();
    // layer 2: wave w owns rows [32 w, 32 w + 32).  The arithmetic of enc_finish_32rows, written for a wave that has its 32 rows to
    // itself: four partial tiles d_q over k in [32 q, 32 q + 32) (lane (m, h) feeds k = 32 q + 16 h + s at step s to both operands),
    // summed ((d0 + d1) + d2) + d3 -- so that a node's h0 is bit for bit what the 32-row kernel and the split-K tail give it
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
    // h0 in accumulator layout: lane = channel n, register i = row (i & 3) + 8 (i >> 2) + 4 h of the wave's 32
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
    // projections: P[row][slot] = sum_k h0[row][k] Wp[slot][k]; lane (m, h) feeds k = 16 h + s at step s
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

// act[M][O] = [ReLU](bias + sum_ks part[ks][M][O]) -- only for encoders deeper than two layers.
__global__ __launch_bounds__(256) void reduce_bias_act_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              float* __restrict__ act, int M, int O, int ks, int relu) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * O) return;
    float v = bias[idx % O];
    for (int s = 0; s < ks; ++s) v += part[(size_t)s * M * O + idx];
    act[idx] = relu ? fmaxf(v, 0.f) : v;
}

// ------------------------------------------------------------------------------------------------------------
// Per-node projection for the NEXT message-passing step.  Called by one whole wave that holds the node's latent
// h[c] in lane c (c < 32, mirrored in lanes 32..63).  Lane o < 48 produces projection slot o:
//   [0,6) P_dst   [8,14) P_src + b_e   [16,48) Q + b_n          (weights transposed in LDS: [c][48])
// With reattach_initial_nodes the input is cat((initial, latent)) -- initial first (models/mpn.py:285).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void project_node(float h_latent, float h_init, bool reatt_n, const float* s_projT,
                                             const float* projb, float* __restrict__ pd_row,
                                             float* __restrict__ psq_row, int lane) {
    const int o = min(lane, kProjOut - 1);
    float acc = projb[o];
    const float* w = s_projT + o;
    if (reatt_n) {
#pragma unroll
        for (int c = 0; c < kH; ++c)
            acc = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_init), c)), acc);
        w += kH * kProjOut;
    }
#pragma unroll
    for (int c = 0; c < kH; ++c)
        acc = fmaf(w[c * kProjOut], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_latent), c)), acc);
    if (lane < kPdStride)
        pd_row[lane] = acc;
    else if (lane < kProjOut)
        psq_row[lane - kPdStride] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// Encoder tail: finishes the previous GEMM layer (sum of split-K partials in fixed order + bias + ReLU), applies
// the last encoder layer F -> 32 (+ReLU), stores h0 and emits the step-1 projections.  One wave per node.
// Replaces the rest of encoder.node_mlp (models/mpn.py:131) and the x[row]/x[col] gathers of step 1.
// ------------------------------------------------------------------------------------------------------------
struct TailParams {
    const float* blob;
    const float* part;
    float* h0;
    float* trace_h;
    float* pd_out;
    float* psq_out;
    int off_prev_b, off_lastWT, off_last_b, off_projwT, off_projb;
    int ks, F, N, has_last, relu_prev, reatt_n, hin, vec_reduce;
    int npw;   // enc_tail_fast_kernel: nodes (= waves that work) per workgroup, 1 / 2 / 4 (0 = 4)
    // graph-plan repair (runs in the extra last workgroup only when the graph was flagged unsorted)
    const long long* ei;
    int* seg_ptr;
    int* col32;
    int* perm;
    int* cursor;
    unsigned* flags;
    const unsigned* blockflags;
    int E;
    DropCfg drop;   // train-mode Dropout (enc_tail_kernel only): p_enc on both encoder layers
};

__global__ __launch_bounds__(256) void enc_tail_kernel(const TailParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_proj = smem;                                    // [hin][48]
    float* s_last = s_proj + p.hin * kProjOut;               // [F][32]   (has_last)
    float* s_row = s_last + (p.has_last ? p.F * kH : 0);     // [4][F]
    float* s_red = s_row + 4 * p.F;                          // [4][64 floats x 4]  (vec_reduce)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ blob = p.blob;
    if (blockIdx.x == gridDim.x - 1) {  // the plan workgroup: fold the per-block findings, repair if needed
        plan_finish(p.ei, p.E, p.N, p.seg_ptr, p.col32, p.perm, p.cursor, p.flags, p.blockflags,
                    reinterpret_cast<unsigned*>(smem));
        return;
    }
    GNNCCA_STAMP(0, 0);
    const int nblk = gridDim.x - 1;
    const int o = lane & 31, half = lane >> 5;
    float* rowbuf = s_row + wave * p.F;
    float* redbuf = s_red + wave * 256;
    // split-K partial sum of one node row: F/4 float4 chunks per row; 64/(F/4) lane groups walk the partials in an
    // interleaved, fixed order with all loads independent (one round trip instead of ks dependent ones)
    const int nchunk = p.vec_reduce ? (p.F >> 2) : 64, groups = 64 / nchunk;
    const int rg = lane / nchunk, rc = lane - rg * nchunk;
    auto partial_sum = [&](int node) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (p.vec_reduce && node < p.N && rg < groups) {
            const float* __restrict__ src = p.part + (size_t)node * p.F + 4 * rc;
            const size_t sstride = (size_t)p.N * p.F;
#pragma unroll 8
            for (int s = rg; s < p.ks; s += groups) a += *reinterpret_cast<const f32x4*>(src + s * sstride);
        }
        return a;
    };
    // Issue order = completion order (vmcnt): weights first (consumed first, by the LDS stage), then biases, then
    // the first node's partials, so that one round trip covers all three.
    const int n4p = p.hin * kProjOut / 4, n4l = p.has_last ? p.F * kH / 4 : 0;
    const f32x4* __restrict__ gp4 = reinterpret_cast<const f32x4*>(blob + p.off_projwT);
    const f32x4* __restrict__ gl4 = reinterpret_cast<const f32x4*>(blob + p.off_lastWT);
    f32x4 rp[3], rl[4];
#pragma unroll
    for (int u = 0; u < 3; ++u) rp[u] = gp4[min(u * 256 + tid, n4p - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) rl[u] = gl4[min(u * 256 + tid, max(n4l - 1, 0))];
    const float last_b = p.has_last ? blob[p.off_last_b + o] : 0.f;
    const float prev_b0 = blob[p.off_prev_b + min(lane, p.F - 1)];
    const float prev_b1 = blob[p.off_prev_b + min(lane + 64, p.F - 1)];
    f32x4 pre = partial_sum(blockIdx.x * 4 + wave);
    GNNCCA_STAMP(0, 1);
    {
        f32x4* lp4 = reinterpret_cast<f32x4*>(s_proj);
        f32x4* ll4 = reinterpret_cast<f32x4*>(s_last);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u * 256 + tid < n4p) lp4[u * 256 + tid] = rp[u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u * 256 + tid < n4l) ll4[u * 256 + tid] = rl[u];
        for (int i = 1024 + tid; i < n4l; i += 256) ll4[i] = gl4[i];  // F > 128: the rest, plainly
    }
    GNNCCA_STAMP(0, 2);
    __syncthreads();
    GNNCCA_STAMP(0, 3);
    for (int grp = blockIdx.x; grp * 4 < p.N; grp += nblk) {
        const int node = grp * 4 + wave;
        const bool active = node < p.N;
        if (active) {
            if (p.vec_reduce) {
                if (rg < groups) *reinterpret_cast<f32x4*>(redbuf + rg * p.F + 4 * rc) = pre;
                __builtin_amdgcn_wave_barrier();
                for (int f = lane; f < p.F; f += 64) {
                    float v = f < 64 ? prev_b0 : (f < 128 ? prev_b1 : blob[p.off_prev_b + f]);
                    for (int gg = 0; gg < groups; ++gg) v += redbuf[gg * p.F + f];
                    v = p.relu_prev ? fmaxf(v, 0.f) : v;
                    if (p.drop.p_enc > 0.f && p.relu_prev)
                        v *= drop_scale(*p.drop.seed, kDropEncNode1, (unsigned long long)node * p.F + f, p.drop.p_enc);
                    rowbuf[f] = v;
                }
            } else {
                for (int f = lane; f < p.F; f += 64) {
                    float v = blob[p.off_prev_b + f];
                    for (int s = 0; s < p.ks; ++s) v += p.part[((size_t)s * p.N + node) * p.F + f];
                    v = p.relu_prev ? fmaxf(v, 0.f) : v;
                    if (p.drop.p_enc > 0.f && p.relu_prev)
                        v *= drop_scale(*p.drop.seed, kDropEncNode1, (unsigned long long)node * p.F + f, p.drop.p_enc);
                    rowbuf[f] = v;
                }
            }
        }
        GNNCCA_STAMP(0, 4);
        pre = partial_sum((grp + nblk) * 4 + wave);  // next node of this wave, in flight during the layer below
        __syncthreads();
        GNNCCA_STAMP(0, 5);
        float hv = 0.f;
        if (active) {
            if (p.has_last) {
                // four independent accumulators and an 8-deep unroll keep 16 LDS reads in flight per lane
                float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
                int f = half;
                for (; f + 14 < p.F; f += 16) {
#pragma unroll
                    for (int u = 0; u < 8; u += 4) {
                        acc0 = fmaf(rowbuf[f + 2 * u + 0], s_last[(f + 2 * u + 0) * kH + o], acc0);
                        acc1 = fmaf(rowbuf[f + 2 * u + 2], s_last[(f + 2 * u + 2) * kH + o], acc1);
                        acc2 = fmaf(rowbuf[f + 2 * u + 4], s_last[(f + 2 * u + 4) * kH + o], acc2);
                        acc3 = fmaf(rowbuf[f + 2 * u + 6], s_last[(f + 2 * u + 6) * kH + o], acc3);
                    }
                }
                for (; f < p.F; f += 2) acc0 = fmaf(rowbuf[f], s_last[f * kH + o], acc0);
                float acc = (acc0 + acc1) + (acc2 + acc3);
                acc += __shfl_xor(acc, 32);
                hv = fmaxf(acc + last_b, 0.f);
                if (p.drop.p_enc > 0.f) hv *= drop_scale(*p.drop.seed, kDropEncNode2, (unsigned long long)node * kH + o, p.drop.p_enc);
            } else {
                hv = rowbuf[o];
            }
            GNNCCA_STAMP(0, 6);
            if (lane < kH) {
                p.h0[(size_t)node * kH + lane] = hv;
                if (p.trace_h) p.trace_h[(size_t)node * kH + lane] = hv;
            }
            project_node(hv, hv, p.reatt_n != 0, s_proj, blob + p.off_projb, p.pd_out + (size_t)node * kPdStride,
                         p.psq_out + (size_t)node * kPsQStride, lane);
            GNNCCA_STAMP(0, 7);
        }
        __syncthreads();
    }
    GNNCCA_STAMP(0, 8);
}


// ------------------------------------------------------------------------------------------------------------
// Encoder tail, register-resident form for the shipped shape (previous layer 128 wide, last layer 128 -> 32, no
// reattach): the same arithmetic as enc_tail_kernel with NO LDS and no barrier.  A wave keeps its share of the two
// small matrices in VGPRs for all the nodes it processes -- lane (o = l & 31, half = l >> 5) holds W2[f][o] for the 64
// values f = 2 i + half, lane o < 48 holds the 32 projection weights W_p[c][o] -- and broadcasts the node's
// activations with v_readlane (SGPR operands) instead of LDS reads: the last layer is 64 x (2 readlane + select + FMA),
// the projection 32 x (readlane + FMA).  rocprofv3 / s_memtime on the LDS form at N = 256: last layer 1.4 us +
// projection 0.7 us + LDS staging and two barriers 1.3 us of a 6.6 us wave lifetime; at N = 65 536 it was LDS-bound.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_tail_fast_kernel(const TailParams p) {
    constexpr int F = 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ blob = p.blob;
    if (blockIdx.x == gridDim.x - 1) {  // the plan workgroup: fold the per-block findings, repair if needed
        __shared__ unsigned smem[1024];
        plan_finish(p.ei, p.E, p.N, p.seg_ptr, p.col32, p.perm, p.cursor, p.flags, p.blockflags, smem);
        return;
    }
    const int nblk = gridDim.x - 1;
    // small graphs: fewer nodes per workgroup, so that the slab reads of the graph's nodes spread over more CUs (a CU takes its
    // 16 KB per node from L2 at a few ten GB/s; the other waves of the workgroup leave at once: no barrier in this kernel)
    const int npw = p.npw > 0 ? p.npw : 4;
    if (wave >= npw) return;
    const int o = lane & 31, half = lane >> 5;
    // split-K partial sum of one node row: lane (rg = half, rc = o) sums the slabs s = rg, rg + 2, ... of columns 4 rc .. 4 rc + 3
    auto partial_sum = [&](int node) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (node < p.N) {
            const float* __restrict__ src = p.part + (size_t)node * F + 4 * o;
            const size_t sstride = (size_t)p.N * F;
#pragma unroll 8
            for (int s = half; s < p.ks; s += 2) a += *reinterpret_cast<const f32x4*>(src + s * sstride);
        }
        return a;
    };
    f32x4 pre = partial_sum(blockIdx.x * npw + wave);
    // the wave's share of the weights, once
    float w2[64], wp[kH];
#pragma unroll
    for (int i = 0; i < 64; ++i) w2[i] = blob[p.off_lastWT + (2 * i + half) * kH + o];
    const int po = min(lane, kProjOut - 1);
#pragma unroll
    for (int c = 0; c < kH; ++c) wp[c] = blob[p.off_projwT + c * kProjOut + po];
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(blob + p.off_prev_b + 4 * o);
    const float last_b = blob[p.off_last_b + o];
    const float proj_b = blob[p.off_projb + po];
    for (int grp = blockIdx.x; grp * npw < p.N; grp += nblk) {
        const int node = grp * npw + wave;
        f32x4 row = pre;
        pre = partial_sum((grp + nblk) * npw + wave);  // next node of this wave, in flight during the arithmetic below
        if (node >= p.N) continue;
        // both halves -> the full split-K sum, + bias, ReLU: lane l holds columns 4 (l & 31) .. + 3 of the 128-wide row
        // (same association as enc_tail_kernel, so that both tails agree bit for bit: (bias + even slabs) + odd slabs)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float other = __shfl_xor(row[u], 32);
            const float v = (b1[u] + (half ? other : row[u])) + (half ? row[u] : other);
            row[u] = p.relu_prev ? fmaxf(v, 0.f) : v;
        }
        // last layer: column f of the row sits in lane f >> 2, component f & 3; half h takes the columns f = 2 i + h into
        // four accumulators by i & 3 (enc_tail_kernel's order)
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float xa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[(2 * i) & 3]), (2 * i) >> 2));
            const float xb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[(2 * i + 1) & 3]), (2 * i + 1) >> 2));
            acc4[i & 3] = fmaf(half ? xb : xa, w2[i], acc4[i & 3]);
        }
        float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        acc += __shfl_xor(acc, 32);
        const float hv = fmaxf(acc + last_b, 0.f);  // lanes o and o + 32 both hold h[o]
        if (lane < kH) p.h0[(size_t)node * kH + lane] = hv;
        // step-1 projections: slot o < 48
        float pr = proj_b;
#pragma unroll
        for (int c = 0; c < kH; ++c) pr = fmaf(wp[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), c)), pr);
        if (lane < kPdStride)
            p.pd_out[(size_t)node * kPdStride + lane] = pr;
        else if (lane < kProjOut)
            p.psq_out[(size_t)node * kPsQStride + lane - kPdStride] = pr;
    }
}



// ------------------------------------------------------------------------------------------------------------
// The rest of encoder.node_mlp for 32 nodes whose first-layer activations h1 = [ReLU](x W1^T + b1) sit in LDS
// ([32][132] floats): layer 2 (128 -> 32) and the step-1 projections (32 -> 48) on v_mfma_f32_32x32x2_f32 (exact
// fp32 FMA chains).  256 threads; the caller's barrier has published s_h1.
//   layer 2 : wave w takes k in [32 w, 32 w + 32) (lane (m, h) feeds k = 32 w + 16 h + s at step s to both
//             operands), the four partial tiles are summed in the fixed order ((d0 + d1) + d2) + d3;
//   project : waves 0 and 1, column tile = wave, lane (m, h) feeds k = 16 h + s.
// ONE definition for the split-K tail (enc_tail_mfma_kernel), the 32-row fused GEMM below and -- as the same
// arithmetic written for a wave that owns its 32 rows alone -- the 256-row GEMM's epilogue: a node's h0 and
// projections are a function of its h1 row only, bit for bit, whichever of the three produced it.
// Replaces the second nn.Linear of encoder.node_mlp (models/mpn.py:131) and the x[row] / x[col] gathers of step 1.
// ------------------------------------------------------------------------------------------------------------
constexpr int kFinLD1 = 132, kFinLDP = 33, kFinLD0 = 36;

struct EncFinishOut {
    const float* W2rm;     // [32][128] row-major [out][in]
    const float* b2;       // [32]
    const float* projwT;   // [32][48] k-major
    const float* projb;    // [48]
    float* h0;             // [N][32]
    float* trace_h;        // [N][32] or null
    float* pd_out;         // [N][8]
    float* psq_out;        // [N][40]
};

// The weights a thread needs in enc_finish_32rows, in registers.  enc_finish_preload issues their loads: a caller that still waits for
// its own operands (the split-K tail: eight slabs from HBM) calls it FIRST, so that the three L2 round trips -- W2 fragments, b2,
// projection weights -- overlap that wait instead of following one barrier each.
struct EncFinishRegs {
    float w2[16];      // W2[l32][32 wave + 16 h + s]
    float b2[4];       // b2[4 (tid & 7) + q]
    float pw[16];      // projwT[16 h + s][32 wave + l32]   (waves 0, 1; 0 beyond slot 47)
    float pb;          // projb[32 wave + l32]
};

__device__ __forceinline__ EncFinishRegs enc_finish_preload(const EncFinishOut& o) {
    constexpr int F = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    EncFinishRegs r;
    const float* wr = o.W2rm + (size_t)l32 * F + 32 * wave + 16 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(wr + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q) r.w2[4 * j + q] = b4[q];
    }
    const f32x4 bb = *reinterpret_cast<const f32x4*>(o.b2 + (tid & 7) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) r.b2[q] = bb[q];
    const int slot = 32 * (wave & 1) + l32;
    const bool on = slot < kProjOut;
#pragma unroll
    for (int s = 0; s < 16; ++s) r.pw[s] = on ? o.projwT[(16 * h + s) * kProjOut + slot] : 0.f;
    r.pb = on ? o.projb[slot] : 0.f;
    return r;
}

__device__ __forceinline__ void enc_finish_32rows(const float* s_h1, float* s_dp, float* s_h0, const EncFinishOut& o, const EncFinishRegs& w,
                                                  int node0, int N) {
    constexpr int LD1 = kFinLD1, LDP = kFinLDP, LD0 = kFinLD0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    {
        const float* hr = s_h1 + l32 * LD1 + 32 * wave + 16 * h;
        float av[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[4 * j + q] = a4[q];
        }
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], w.w2[s], d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) s_dp[(wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * LDP + l32] = d[i];
    }
    __syncthreads();
    {
        const int r = tid >> 3, c4 = (tid & 7) * 4;
        const int node = node0 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c4 + q;
            float v = s_dp[r * LDP + c];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) v += s_dp[(ww * 32 + r) * LDP + c];
            v = fmaxf(v + w.b2[q], 0.f);
            s_h0[r * LD0 + c] = v;
            if (node < N) {
                o.h0[(size_t)node * kH + c] = v;
                if (o.trace_h) o.trace_h[(size_t)node * kH + c] = v;
            }
        }
    }
    __syncthreads();
    if (wave < 2) {
        const int slot = 32 * wave + l32;
        const bool on = slot < kProjOut;
        float a2[16];
        const float* hr = s_h0 + l32 * LD0 + 16 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(hr + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) a2[4 * j + q] = a4[q];
        }
        f32x16 pacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], w.pw[s], pacc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int node = node0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (node < N && on) {
                const float v = pacc[i] + w.pb;
                if (slot < kPdStride)
                    o.pd_out[(size_t)node * kPdStride + slot] = v;
                else
                    o.psq_out[(size_t)node * kPsQStride + slot - kPdStride] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Encoder tail on the matrix pipe, for batches whose first GEMM ran split-K (4096 <= N, partial slabs in HBM): a
// 256-thread workgroup finishes 32 nodes.
//   1. every thread sums its 16 columns of one node over the ks slabs (whole 512-B rows, four 16-B loads per slab, all
//      slabs in flight), adds the bias, applies ReLU and parks h1 [32][128] in LDS;
//   2. layer 2 (128 -> 32) as v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains): wave w takes k in [32 w, 32 w + 32) -- the
//      A operand straight from LDS (row stride 132 floats: ds_read_b128 of 16 rows hits 64 distinct banks), the B operand
//      straight from the row-major W2 in the blob -- and the four partial tiles are summed in fixed order;
//   3. the step-1 projections (32 -> 48) the same way by waves 0 and 1.
// The wave-per-node VALU + LDS form it replaces took 15 us at N = 8 192 and 22 us at N = 16 384 (63 us at 65 536, where the
// GEMM's fused epilogue now does this work).  Same plan workgroup as the other tails.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_tail_mfma_kernel(const TailParams p, const float* __restrict__ W2rm) {
    constexpr int F = 128, LD1 = 132, LDP = 33, LD0 = 36;
    __shared__ __attribute__((aligned(16))) float s_h1[32 * LD1];
    __shared__ __attribute__((aligned(16))) float s_dp[4 * 32 * LDP];
    __shared__ __attribute__((aligned(16))) float s_h0[32 * LD0];
    const int tid = threadIdx.x;
    const float* __restrict__ blob = p.blob;
    if (blockIdx.x == gridDim.x - 1) {  // the plan workgroup: fold the per-block findings, repair if needed
        __shared__ unsigned smem[1024];
        plan_finish(p.ei, p.E, p.N, p.seg_ptr, p.col32, p.perm, p.cursor, p.flags, p.blockflags, smem);
        return;
    }
    const int node0 = blockIdx.x * 32;
    EncFinishOut fo;
    fo.W2rm = W2rm, fo.b2 = blob + p.off_last_b, fo.projwT = blob + p.off_projwT, fo.projb = blob + p.off_projb;
    fo.h0 = p.h0, fo.trace_h = p.trace_h, fo.pd_out = p.pd_out, fo.psq_out = p.psq_out;
    const EncFinishRegs fw = enc_finish_preload(fo);   // the weights of steps 2 and 3, requested before the slabs
    // ---- 1. slab sum + bias + ReLU -> LDS -------------------------------------------------------------------------------------
    {
        // thread (node nl, t = tid & 7) owns the columns 32 j + 4 t .. + 3, j = 0..3: for a fixed j the eight threads of a row read
        // one whole 128-B line (16 columns in a row per thread would touch every line of the row in every instruction)
        const int nl = tid >> 3, c0 = (tid & 7) * 4;
        const int node = min(node0 + nl, p.N - 1);
        const float* __restrict__ src = p.part + (size_t)node * F + c0;
        const size_t sstride = (size_t)p.N * F;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(blob + p.off_prev_b + c0 + 32 * j);
        for (int s0 = 0; s0 < p.ks; s0 += 8) {   // eight slabs (32 loads) in flight, summed in slab order
            f32x4 t[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[u][j] = (s0 + u < p.ks) ? *reinterpret_cast<const f32x4*>(src + (size_t)(s0 + u) * sstride + 32 * j)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += t[u][j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (p.relu_prev) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][q] = fmaxf(v[j][q], 0.f);
            }
            *reinterpret_cast<f32x4*>(s_h1 + nl * LD1 + c0 + 32 * j) = v[j];
        }
    }
    __syncthreads();
    // ---- 2. / 3. layer 2 with k split over the four waves, projections by waves 0 and 1 (enc_finish_32rows) -------------------------
    enc_finish_32rows(s_h1, s_dp, s_h0, fo, fw, node0, p.N);
}


}  // namespace gnncca
