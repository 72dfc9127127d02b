// How many cycles does a v_mfma_f32_32x32x16_bf16 cost a wave when consecutive MFMAs accumulate into the SAME tile (a dependent chain,
// the inner loop of the split-bf16 encoder GEMM: six products per accumulator) vs round-robin over 2 / 4 independent tiles, with one
// (chains of CH MFMAs per tile before moving on), with one and with two waves per SIMD?  s_memtime around 4096 MFMAs per wave.  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int CH>
__global__ __launch_bounds__(1024) void k_chain(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 av, bv;
#pragma unroll
    for (int q = 0; q < 8; ++q) av[q] = (__bf16)(float)(threadIdx.x & 7), bv[q] = (__bf16)(float)(q + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            constexpr int dummy = 0;
            (void)dummy;
            acc[(u / CH) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[(u / CH) % NACC], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][15];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int CH>
void run(int waves_per_cu, unsigned long long* out, float* sink) {
    const int iters = 170;   // 24 * 170 = 4080 MFMAs per wave
    hipLaunchKernelGGL((k_chain<NACC, CH>), dim3(256), dim3(64 * waves_per_cu), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((k_chain<NACC, CH>), dim3(256), dim3(64 * waves_per_cu), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, out, sizeof(unsigned long long) * 256 * waves_per_cu, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 256 * waves_per_cu; ++i) sum += (double)h[i];
    const double per = sum / (256.0 * waves_per_cu) / (24.0 * iters);
    printf("accumulators %d, chain %2d, waves per SIMD %d: %.1f cycles per MFMA per wave -> %.1f pipe cycles per MFMA on the SIMD\n", NACC, CH, waves_per_cu / 4,
           per, per / (waves_per_cu / 4));
}

int main() {
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, sizeof(unsigned long long) * 4096);
    hipMalloc(&sink, 64);
    for (int w : {4, 8, 16}) {
        run<1, 1>(w, out, sink);
        run<4, 1>(w, out, sink);
        run<4, 2>(w, out, sink);
        run<4, 3>(w, out, sink);
        run<4, 6>(w, out, sink);
        run<2, 12>(w, out, sink);
        run<4, 12>(w, out, sink);   // 24 MFMAs per trip: only two of the four tiles are touched
        run<2, 1>(w, out, sink);
        run<2, 6>(w, out, sink);
        run<3, 8>(w, out, sink);
        run<4, 24>(w, out, sink);   // one tile per trip, four tiles in rotation over trips
    }
    return 0;
}
