// Round 5: how fast can a 512-thread workgroup per CU stream its 256 rows x 2048 floats when every WAVE owns 32 rows and
// fills a private LDS ring by LDS-DMA (buffer_load_dwordx4 ... lds: 8 rows x 128 B per wave-instruction, no VGPR staging, no
// ds_write), D chunks ahead, reading each chunk back with ds_read_b128 -- the x path of the fp16-split encoder GEMM without
// its MFMAs?  Variants: bytes per row and chunk (CB), ring depth (S), a per-workgroup rotation of the k order (ROT), a shared
// W ring filled the same way with one workgroup barrier per chunk (WRING).  Read-only; a checksum keeps the reads alive.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_xring.hip -o tools/ubench_xring.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int CB, int S, bool ROT, bool WRING>
__device__ __forceinline__ void ring_body(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ sink, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int XSLOT = 32 * CB;            // bytes per wave per slot
    constexpr int NI = XSLOT / 1024;          // wave-instructions per chunk (1 KB each)
    constexpr int RPI = 1024 / CB;            // rows per wave-instruction
    constexpr int WSLOT = WRING ? 2 * 128 * (CB / 2) : 0;   // fp16 pieces: 2 x 128 cols x (CB/4 k) x 2 B
    constexpr int WI = WSLOT / 1024 / 8;      // W wave-instructions per wave per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K * 4 / CB;
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)M * K * 4), 0x00020000);
    const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)((size_t)K * 128 * 2 * 2), 0x00020000);
    unsigned char* xring = lds + (size_t)wave * S * XSLOT;
    unsigned char* wring = lds + (size_t)8 * S * XSLOT;
    const int row0 = blockIdx.x * 256 + wave * 32;
    unsigned voff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int rl = j * RPI + lane / (CB / 16), g = lane % (CB / 16);
        const int gs = g ^ ((rl >> 1) & 7);   // source granule (16 B) that lands in LDS slot g of row rl
        voff[j] = (unsigned)(((size_t)(row0 + rl) * K) * 4 + gs * 16);
    }
    const int rot = ROT ? (int)((blockIdx.x * 37u) % (unsigned)nk) : 0;
    auto issue = [&](int kt) {
        const int kc = (kt + rot) % nk;
        const int slot = kt % S;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xring + slot * XSLOT + j * 1024), 16, voff[j], kc * CB, 0, 0);
        if (WRING) {
#pragma unroll
            for (int j = 0; j < WI; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wring + slot * WSLOT + (wave * WI + j) * 1024), 16, lane * 16,
                                                         kc * WSLOT + (wave * WI + j) * 1024, 0, 0);
        }
    };
    constexpr int D = S - 1;
    constexpr int PER = NI + WI;   // DMA instructions per chunk per wave
    for (int d = 0; d < D; ++d) issue(d);
    f4 acc = {0, 0, 0, 0};
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + D < nk) {
            issue(kt + D);
            if (D == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            if (D == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            if (D == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (WRING) __builtin_amdgcn_s_barrier();   // the chunk's W piece is every wave's: visible after all waves waited for theirs
        const int slot = kt % S;
        const f4* xs = reinterpret_cast<const f4*>(xring + slot * XSLOT);
        // fragment-shaped read back: lane (r, h) reads 32 B per 16-deep k-step of its row
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int s = 0; s < CB / 64; ++s) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = (4 * s + 2 * h + q) ^ ((r >> 1) & 7);
                acc += xs[r * (CB / 16) + (g % (CB / 16))];
            }
        }
        if (WRING) {
            const f4* ws = reinterpret_cast<const f4*>(wring + slot * WSLOT);
#pragma unroll
            for (int q = 0; q < WSLOT / 1024 / 2; ++q) acc += ws[q * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave is done with this W slot before the next iteration's DMA overwrites slot (kt + D + 1) % S ... (kt+S)%S
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// (concrete kernels: with this toolchain a __global__ TEMPLATE whose body issues the LDS-DMA builtin from a lambda is not emitted)
#define DEF(name, CB, S, ROT, WR) \
    __global__ __launch_bounds__(512) void name(const float* x, const float* w, float* sink, int M, int K) { ring_body<CB, S, ROT, WR>(x, w, sink, M, K); }
DEF(k_128_2, 128, 2, false, false)
DEF(k_128_3, 128, 3, false, false)
DEF(k_128_4, 128, 4, false, false)
DEF(k_128_4r, 128, 4, true, false)
DEF(k_128_3r, 128, 3, true, false)
DEF(k_256_2, 256, 2, false, false)
DEF(k_256_2r, 256, 2, true, false)
DEF(k_128_3w, 128, 3, false, true)
DEF(k_128_3rw, 128, 3, true, true)
DEF(k_128_2w, 128, 2, false, true)

static float *g_x, *g_w, *g_sink;
static hipEvent_t g_e0, g_e1;
constexpr int N = 65536, K = 2048;

typedef void (*kern_t)(const float*, const float*, float*, int, int);
static void run(kern_t kern, int CB, int S, bool WR, const char* name) {
    const int ldsb = 8 * S * 32 * CB + (WR ? S * 2 * 128 * (CB / 2) : 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(N / 256), dim3(512), ldsb, 0, g_x, g_w, g_sink, N, K);
    (void)hipEventRecord(g_e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(N / 256), dim3(512), ldsb, 0, g_x, g_w, g_sink, N, K);
    (void)hipEventRecord(g_e1);
    (void)hipEventSynchronize(g_e1);
    float ms;
    (void)hipEventElapsedTime(&ms, g_e0, g_e1);
    printf("%-48s %8.1f us  %6.2f TB/s  lds %d (%s)\n", name, ms * 100, (double)N * K * 4 / (ms / 10 * 1e-3) / 1e12, ldsb, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&g_x, (size_t)N * K * 4);
    (void)hipMalloc(&g_w, (size_t)K * 128 * 4);
    (void)hipMalloc(&g_sink, 64);
    (void)hipMemset(g_x, 0, (size_t)N * K * 4);
    (void)hipMemset(g_w, 0, (size_t)K * 128 * 4);
    (void)hipEventCreate(&g_e0);
    (void)hipEventCreate(&g_e1);
    run(k_128_2, 128, 2, false, "x ring 128 B/row, 2 slots (1 ahead)");
    run(k_128_3, 128, 3, false, "x ring 128 B/row, 3 slots (2 ahead)");
    run(k_128_4, 128, 4, false, "x ring 128 B/row, 4 slots (3 ahead)");
    run(k_128_4r, 128, 4, false, "x ring 128 B/row, 4 slots, rotated k");
    run(k_128_3r, 128, 3, false, "x ring 128 B/row, 3 slots, rotated k");
    run(k_256_2, 256, 2, false, "x ring 256 B/row, 2 slots (1 ahead)");
    run(k_256_2r, 256, 2, false, "x ring 256 B/row, 2 slots, rotated k");
    run(k_128_3w, 128, 3, true, "x + W rings 128 B/row, 3 slots, barriers");
    run(k_128_3rw, 128, 3, true, "x + W rings 128 B/row, 3 slots, rot, barriers");
    run(k_128_2w, 128, 2, true, "x + W rings 128 B/row, 2 slots, barriers");
    return 0;
}
