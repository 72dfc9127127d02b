// How fast can a 512-thread workgroup per CU stream its 256-row x 2048-float tile of a row-major fp32 matrix when it walks
// the tile CHUNK by chunk (all 256 rows x CB bytes, then the next CB bytes of every row) with a barrier per chunk -- the
// access pattern of the split-bf16 encoder GEMM -- for CB = 128, 256, 512 B per row and chunk, vs a plain linear sweep of the
// same 2 MB?  Read-only; a checksum keeps the loads alive.  hipcc --offload-arch=gfx950 -O3 tools/ubench_rowtile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CB, int ROWS = 256, int THREADS = 512>   // bytes per row per chunk; rows per workgroup; threads per workgroup
__global__ __launch_bounds__(THREADS) void k_tile(const float* __restrict__ x, float* __restrict__ sink, int K, int deep) {
    constexpr int F4 = CB / 16;            // float4 per row per chunk
    constexpr int PER = ROWS * F4 / THREADS;    // float4 per thread per chunk
    extern __shared__ float dyn_lds[];   // only to limit residency (launch with e.g. 96 KB -> one workgroup per CU)
    const int tid = threadIdx.x;
    const float* base = x + (size_t)blockIdx.x * ROWS * K;
    f4 acc = {0, 0, 0, 0};
    if (deep == 12345) dyn_lds[tid] = 1.f;
    const int nchunk = K * 4 / CB;
    f4 r[2][PER];
    auto load = [&](int kt, f4 (&dst)[PER]) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + THREADS * u, row = q / F4, c = q % F4;
            dst[u] = *reinterpret_cast<const f4*>(base + (size_t)row * K + kt * (CB / 4) + c * 4);
        }
    };
    load(0, r[0]);
    for (int kt = 0; kt < nchunk; kt += 2) {
        if (kt + 1 < nchunk) load(kt + 1, r[1]);
#pragma unroll
        for (int u = 0; u < PER; ++u) acc += r[0][u];
        __syncthreads();
        if (kt + 2 < nchunk) load(kt + 2, r[0]);
#pragma unroll
        for (int u = 0; u < PER; ++u) acc += r[1][u];
        __syncthreads();
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

__global__ __launch_bounds__(512) void k_linear(const float* __restrict__ x, float* __restrict__ sink, int K) {
    const f4* base = reinterpret_cast<const f4*>(x + (size_t)blockIdx.x * 256 * K);
    f4 acc = {0, 0, 0, 0};
    const int n4 = 256 * K / 4;
    for (int i = threadIdx.x; i < n4; i += 512 * 4) {
        f4 a = base[i], b = base[min(i + 512, n4 - 1)], c = base[min(i + 1024, n4 - 1)], d = base[min(i + 1536, n4 - 1)];
        acc += a + b + c + d;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const int N = 65536, K = 2048;
    float *x, *sink;
    hipMalloc(&x, (size_t)N * K * 4);
    hipMalloc(&sink, 64);
    hipMemset(x, 0, (size_t)N * K * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms * 100, (double)N * K * 4 / (ms / 10 * 1e-3) / 1e12);
    };
    time("tile chunk 128 B/row", [&] { hipLaunchKernelGGL(k_tile<128>, dim3(N / 256), dim3(512), 0, 0, x, sink, K, 0); });
    time("tile chunk 256 B/row", [&] { hipLaunchKernelGGL(k_tile<256>, dim3(N / 256), dim3(512), 0, 0, x, sink, K, 0); });
    time("tile chunk 512 B/row", [&] { hipLaunchKernelGGL(k_tile<512>, dim3(N / 256), dim3(512), 0, 0, x, sink, K, 0); });
    time("tile chunk 1024 B/row", [&] { hipLaunchKernelGGL(k_tile<1024>, dim3(N / 256), dim3(512), 0, 0, x, sink, K, 0); });
    time("linear 2 MB per workgroup", [&] { hipLaunchKernelGGL(k_linear, dim3(N / 256), dim3(512), 0, 0, x, sink, K); });
    // more, smaller workgroups per CU
    time("128 rows x 128 B, 256 thr", [&] { hipLaunchKernelGGL((k_tile<128, 128, 256>), dim3(N / 128), dim3(256), 0, 0, x, sink, K, 0); });
    time("64 rows x 128 B, 256 thr", [&] { hipLaunchKernelGGL((k_tile<128, 64, 256>), dim3(N / 64), dim3(256), 0, 0, x, sink, K, 0); });
    time("64 rows x 256 B, 256 thr", [&] { hipLaunchKernelGGL((k_tile<256, 64, 256>), dim3(N / 64), dim3(256), 0, 0, x, sink, K, 0); });
    time("32 rows x 512 B, 256 thr", [&] { hipLaunchKernelGGL((k_tile<512, 32, 256>), dim3(N / 32), dim3(256), 0, 0, x, sink, K, 0); });
    time("128 rows x 256 B, 512 thr", [&] { hipLaunchKernelGGL((k_tile<256, 128, 512>), dim3(N / 128), dim3(512), 0, 0, x, sink, K, 0); });
    // the same shapes with ONE workgroup per CU (96 KB of dynamic LDS requested), as a GEMM whose stages fill the LDS would run
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<512, 64, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<256, 128, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<512, 128, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<512, 32, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    time("64 rows x 512 B, 256 thr, 1 WG/CU", [&] { hipLaunchKernelGGL((k_tile<512, 64, 256>), dim3(N / 64), dim3(256), 98304, 0, x, sink, K, 0); });
    time("128 rows x 256 B, 512 thr, 1 WG/CU", [&] { hipLaunchKernelGGL((k_tile<256, 128, 512>), dim3(N / 128), dim3(512), 98304, 0, x, sink, K, 0); });
    time("128 rows x 512 B, 512 thr, 1 WG/CU", [&] { hipLaunchKernelGGL((k_tile<512, 128, 512>), dim3(N / 128), dim3(512), 98304, 0, x, sink, K, 0); });
    time("32 rows x 512 B, 256 thr, 1 WG/CU", [&] { hipLaunchKernelGGL((k_tile<512, 32, 256>), dim3(N / 32), dim3(256), 98304, 0, x, sink, K, 0); });
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<512, 64, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    time("64 rows x 512 B, 256 thr, 2 WG/CU", [&] { hipLaunchKernelGGL((k_tile<512, 64, 256>), dim3(N / 64), dim3(256), 65536, 0, x, sink, K, 0); });
    return 0;
}
