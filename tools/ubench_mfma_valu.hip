// Do the matrix pipe and the VALU of one SIMD overlap ACROSS waves, and at what price?  (round 3, step-kernel diagnosis)
// One workgroup per CU; waves are dealt round-robin to the four SIMDs, so with 512 threads every SIMD holds two waves
// (wave w and w + 4).  Roles per wave: M = a loop of MFMAs, V = a loop of independent v_fma_f32, I = idle (exits).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o tools/ubench_mfma_valu.bin && tools/ubench_mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// role codes: 0 idle, 1 MFMA f32 32x32x2, 2 MFMA bf16 32x32x16, 3 VALU v_fma_f32, 4 VALU v_pk_fma_f32, 5 MFMA f32 two independent accumulators
template <int R0, int R1>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const int role = (wave < 4) ? R0 : R1;
    float acc = threadIdx.x * 1e-9f;
    if (role == 1 || role == 5) {
        f32x16 d = {0}, e = {0};
        const float a = acc + 1.f, b = acc + 2.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
                if (role == 5) e = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, e, 0, 0, 0);
            }
        }
        acc = d[0] + d[7] + e[3];
    } else if (role == 2) {
        f32x16 d = {0};
        bf16x8 a, b;
        for (int q = 0; q < 8; ++q) a[q] = (__bf16)(acc + q), b[q] = (__bf16)(acc - q);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
        }
        acc = d[0] + d[7];
    } else if (role == 3) {
        float x[8];
        for (int q = 0; q < 8; ++q) x[q] = acc + q;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(acc));
        }
        for (int q = 0; q < 8; ++q) acc += x[q];
    } else if (role == 4) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 x[8];
        for (int q = 0; q < 8; ++q) x[q] = f32x2{acc + q, acc - q};
        const f32x2 c = {acc, acc};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(c));
        }
        for (int q = 0; q < 8; ++q) acc += x[q][0] + x[q][1];
    } else {
        return;
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int R0, int R1>
static float run(const char* name, float* out, int iters, int per_iter0, int per_iter1) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<R0, R1>), dim3(256), dim3(512), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<R0, R1>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ns_per_iter = ms * 1e6 / iters;
    printf("%-44s %8.3f ms  %7.1f ns/iter", name, ms, ns_per_iter);
    if (per_iter0) printf("  | waves 0-3: %5.2f ns per op", ns_per_iter / per_iter0);
    if (per_iter1) printf("  | waves 4-7: %5.2f ns per op", ns_per_iter / per_iter1);
    printf("\n");
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int it = 20000;
    run<1, 0>("MFMA f32 32x32x2 (1 wave/SIMD, 8/iter)", out, it, 8, 0);
    run<5, 0>("MFMA f32 32x32x2, two chains (16/iter)", out, it, 16, 0);
    run<1, 1>("MFMA f32 + MFMA f32 (2 waves/SIMD)", out, it, 8, 8);
    run<2, 0>("MFMA bf16 32x32x16 (8/iter)", out, it, 8, 0);
    run<3, 0>("v_fma_f32 (1 wave/SIMD, 32/iter)", out, it, 32, 0);
    run<3, 3>("v_fma_f32 + v_fma_f32 (2 waves/SIMD)", out, it, 32, 32);
    run<4, 0>("v_pk_fma_f32 (32/iter)", out, it, 32, 0);
    run<4, 4>("v_pk_fma_f32 + v_pk_fma_f32", out, it, 32, 32);
    run<1, 3>("MFMA f32 (8/iter) || v_fma_f32 (32/iter)", out, it, 8, 32);
    run<1, 4>("MFMA f32 (8/iter) || v_pk_fma_f32 (32/iter)", out, it, 8, 32);
    run<2, 3>("MFMA bf16 (8/iter) || v_fma_f32 (32/iter)", out, it, 8, 32);
    run<2, 4>("MFMA bf16 (8/iter) || v_pk_fma_f32 (32/iter)", out, it, 8, 32);
    return 0;
}
