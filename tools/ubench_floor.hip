// Micro-benchmark: what does a dependent same-stream launch cost on this box, as a function of grid size and of
// a minimal memory round trip?  (rocprofv3 --kernel-trace --stats gives the per-kernel durations.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_copy6(const float* __restrict__ a, float* __restrict__ b, int n, int stride) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) for (int f = 0; f < 6; ++f) b[f * stride + i] = a[f * stride + i] * 1.5f;
}
__global__ void k_chain3(const int* __restrict__ idx, const float* __restrict__ t, const float* __restrict__ a, float* __restrict__ b, int n, int stride) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { int s = idx[blockIdx.x]; int j = idx[(s + i) % n]; float v = t[j % 2048];
        for (int f = 0; f < 6; ++f) b[f * stride + i] = a[f * stride + i] + v; }
}
int main() {
    const int n = 65280, stride = 65536;
    float *a, *b, *t; int* idx;
    hipMalloc(&a, 6 * stride * 4); hipMalloc(&b, 6 * stride * 4); hipMalloc(&t, 2048 * 4); hipMalloc(&idx, stride * 4);
    hipMemset(a, 0, 6 * stride * 4); hipMemset(idx, 0, stride * 4); hipMemset(t, 0, 2048 * 4);
    hipStream_t st; hipStreamCreate(&st);
    for (int it = 0; it < 200; ++it) {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, (int*)a);
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (int*)a);
        hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, (int*)a);
        hipLaunchKernelGGL(k_copy6, dim3(256), dim3(256), 0, st, a, b, n, stride);
        hipLaunchKernelGGL(k_chain3, dim3(256), dim3(256), 0, st, idx, t, b, a, n, stride);
    }
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int it = 0; it < 1000; ++it) hipLaunchKernelGGL(k_copy6, dim3(256), dim3(256), 0, st, a, b, n, stride);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("1000 dependent k_copy6 launches: %.2f us each (wall, incl. boundaries)\n", ms);
    hipEventRecord(e0, st);
    for (int it = 0; it < 1000; ++it) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (int*)a);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
    printf("1000 dependent empty 256-block launches: %.2f us each\n", ms);
    return 0;
}
