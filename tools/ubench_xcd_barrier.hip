// Price of a grid barrier among the 32 workgroups of ONE XCD (proceed with an in-launch form of the
// single-graph MPN only if this is <= 1.2 us).  256 workgroups are launched (dealt round-robin to the 8 XCDs); only those with
// blockIdx % 8 == 0 take part, so all arrivals and polls meet in one L2.  Every spin is BOUNDED: a participant that does not see
// the others within ~2 ms gives up and raises a flag, the kernel always drains.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_xcd_barrier.hip -o tools/ubench_xcd_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int STRIDE, int SLEEP, bool TREE>   // participants: blockIdx % STRIDE == 0; SLEEP: s_sleep argument between polls; TREE: 8-wide fan-in
__global__ __launch_bounds__(256) void k(unsigned* counter, unsigned* fail, unsigned long long* cycles, int rounds, float* sink) {
    if (blockIdx.x % STRIDE != 0) return;
    const unsigned parts = (gridDim.x + STRIDE - 1) / STRIDE;
    float acc = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.f;          // a token of work between barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned want = parts * (unsigned)(r + 1);
            if (TREE) {   // groups of 8 arrive on their own cache line; the last of a group arrives at the top counter
                const unsigned me = blockIdx.x / STRIDE, grp = me / 8, ngrp = (parts + 7) / 8;
                const unsigned gsize = min(8u, parts - grp * 8);
                const unsigned old = __hip_atomic_fetch_add(counter + 64 * (1 + grp), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == gsize * (unsigned)(r + 1)) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                want = ngrp * (unsigned)(r + 1);
            } else {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > 200000) {     // ~2 ms: never hang the box
                    __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 123.f) *sink = acc;
}

template <int STRIDE, int SLEEP, bool TREE>
static void run(const char* name, int grid, int rounds) {
    unsigned *counter, *fail;
    unsigned long long* cycles;
    float* sink;
    hipMalloc(&counter, 64 * 1024);
    hipMalloc(&fail, 256);
    hipMalloc(&cycles, 8 * 1024);
    hipMalloc(&sink, 256);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(counter, 0, 64 * 1024);
        hipMemset(fail, 0, 256);
        hipMemset(cycles, 0, 8 * 1024);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<STRIDE, SLEEP, TREE>), dim3(grid), dim3(256), 0, 0, counter, fail, cycles, rounds, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned f = 0;
        hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        unsigned long long c0 = 0;
        hipMemcpy(&c0, cycles, 8, hipMemcpyDeviceToHost);
        if (rep == 1)
            printf("%-52s %8.3f ms for %d barriers = %6.2f us each (%llu cycles each on block 0)%s\n", name, ms, rounds, ms * 1e3 / rounds,
                   c0 / (unsigned long long)rounds, f ? "  ** GAVE UP (participants not co-resident?) **" : "");
    }
}

int main() {
    run<8, 1, false>("32 WGs of one XCD, s_sleep 1", 256, 2000);
    run<8, 0, false>("32 WGs of one XCD, no sleep", 256, 2000);
    run<8, 8, false>("32 WGs of one XCD, s_sleep 8", 256, 2000);
    run<8, 1, true>("32 WGs of one XCD, 8-wide tree, s_sleep 1", 256, 2000);
    run<8, 4, true>("32 WGs of one XCD, 8-wide tree, s_sleep 4", 256, 2000);
    run<1, 1, false>("256 WGs, all XCDs, s_sleep 1", 256, 2000);
    run<1, 1, false>("8 WGs, one per XCD", 8, 2000);
    run<8, 1, false>("4 WGs of one XCD (of 32 launched)", 32, 2000);
    run<8, 1, false>("8 WGs of one XCD (of 64 launched)", 64, 2000);
    run<8, 1, false>("16 WGs of one XCD (of 128 launched)", 128, 2000);
    return 0;
}
