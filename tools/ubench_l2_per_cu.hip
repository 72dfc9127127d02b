// What does ONE CU get out of L2 through its vector L1 when every workgroup re-reads the same 1.5 MB table (the W pieces of the
// un-split 32-row encoder GEMM, csrc/enc_rows32.cuh)?  G workgroups of 256 threads, one per CU (a dynamic-LDS request keeps them
// apart), each sweeps the table REPS times with D 16-byte loads in flight per lane, contiguous per wave instruction.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_per_cu.hip -o tools/ubench_l2_per_cu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void k_sweep(const f4* __restrict__ tab, float* __restrict__ sink, int n4, int reps) {
    extern __shared__ float dyn_lds[];
    if (reps == 12345) dyn_lds[threadIdx.x] = 1.f;
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n4; i += 256 * D) {
            f4 v[D];
#pragma unroll
            for (int u = 0; u < D; ++u) v[u] = tab[min(i + 256 * u, n4 - 1)];
#pragma unroll
            for (int u = 0; u < D; ++u) acc += v[u];
        }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int D>
static void run(const f4* tab, float* sink, int n4, int groups, int waves_lds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int reps = 4;
    const size_t lds = waves_lds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_sweep<D>, dim3(groups), dim3(256), lds, 0, tab, sink, n4, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_sweep<D>, dim3(groups), dim3(256), lds, 0, tab, sink, n4, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n4 * 16 * reps;
    printf("D=%2d loads in flight per lane, %3d workgroups (%s per CU): %7.1f us per sweep of %.2f MB -> %6.1f GB/s per workgroup, %6.2f TB/s chip\n",
           D, groups, lds > 80 * 1024 ? "one" : "two+", ms * 1e3 / reps, n4 * 16 / 1e6, bytes / (ms * 1e-3) / 1e9,
           bytes * groups / (ms * 1e-3) / 1e12);
}

int main() {
    const int n4 = 1536 * 1024 / 16;   // 1.5 MB
    f4* tab;
    float* sink;
    hipMalloc(&tab, (size_t)n4 * 16);
    hipMalloc(&sink, 64);
    hipMemset(tab, 0, (size_t)n4 * 16);
    for (int groups : {64, 128, 256}) {
        run<4>(tab, sink, n4, groups, 100 * 1024);
        run<8>(tab, sink, n4, groups, 100 * 1024);
        run<16>(tab, sink, n4, groups, 100 * 1024);
    }
    run<8>(tab, sink, n4, 512, 60 * 1024);    // two workgroups per CU
    run<16>(tab, sink, n4, 512, 60 * 1024);
    run<16>(tab, sink, n4, 1024, 36 * 1024);  // four per CU
    return 0;
}
