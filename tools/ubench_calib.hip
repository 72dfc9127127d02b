// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access pattern: plane-wise
// dword-per-lane streaming loads/stores (256 contiguous bytes per wave instruction), far beyond the 256 MB
// Infinity Cache.  Known byte counts: k_planes reads 6*n*4 B and writes 6*n*4 B.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_planes(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) for (int f = 0; f < 6; ++f) b[f * n + i] = a[f * n + i] * 1.5f;
}
__global__ void k_vec4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 v = a[i]; v.x *= 1.5f; b[i] = v; }
}
int main() {
    const size_t n = 32u << 20;  // 32 Mi floats per plane -> 768 MiB read + 768 MiB written
    float *a, *b;
    hipMalloc(&a, 6 * n * 4); hipMalloc(&b, 6 * n * 4);
    hipMemset(a, 0, 6 * n * 4);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(k_planes, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(k_vec4, dim3((6 * n / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, 6 * n / 4);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: read %zu write %zu\n", 6 * n * 4, 6 * n * 4);
    return 0;
}
