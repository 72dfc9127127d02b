// Can an LDS-DMA ring feed the step kernel's streams faster than per-wave register loads?  The access pattern of
// mpn_step_fast_kernel on a big batch, stripped of its arithmetic: 6 fp32 feature planes + 1 int32 plane read, 6 planes written
// back in place, 256-slot items.  Three forms over the same bytes:
//   ring   : persistent workgroups of 1 loader wave (global_load_lds_dwordx4 into an R-deep ring of 7-KB slots, counted vmcnt,
//            one raw s_barrier per item) + 4 consumer waves (ds_read, trivial arithmetic, global stores);
//   direct : persistent workgroups of 4 waves, every wave loads its 64 slots of the 7 planes into registers, two items in flight;
//   wgitem : one 256-thread workgroup per item (the shape of the shipped kernel at one wave per 64 slots).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_ring.hip -o tools/ubench_ring.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int SLOT = 7 * 256;   // floats per ring slot

__device__ __forceinline__ float work(float v, int c) { return v * 1.5f + (float)(c & 7); }

template <int R, bool NT>
__global__ __launch_bounds__(320) void k_ring(float* __restrict__ e, const int* __restrict__ col, long long es, int items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, g = blockIdx.x;
    const int i0 = (int)((long long)items * g / G), i1 = (int)((long long)items * (g + 1) / G);
    const int n = i1 - i0;
    if (wave == 4) {
        auto issue = [&](int it) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_off(smem + (it % R) * SLOT));
            const size_t slot0 = (size_t)(i0 + it) * 256 + lane * 4;
#pragma unroll
            for (int f = 0; f < 6; ++f) glds16<NT>(e + (size_t)f * es + slot0, dst + f * 1024);
            glds16<NT>(col + slot0, dst + 6 * 1024);
        };
        for (int it = 0; it < R - 1 && it < n; ++it) issue(it);
        for (int it = 0; it < n; ++it) {
            if (it + R - 2 <= n - 1)
                WAIT_VM(7 * (R - 2));
            else
                WAIT_VM(0);
            BARRIER();
            if (it + R - 1 < n) issue(it + R - 1);
        }
    } else {
        for (int it = 0; it < n; ++it) {
            BARRIER();
            const float* s = smem + (it % R) * SLOT + wave * 64 + lane;
            float v[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) v[f] = s[f * 256];
            const int c = reinterpret_cast<const int*>(s)[6 * 256];
            const size_t slot = (size_t)(i0 + it) * 256 + wave * 64 + lane;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                if (NT)
                    __builtin_nontemporal_store(work(v[f], c), e + (size_t)f * es + slot);
                else
                    e[(size_t)f * es + slot] = work(v[f], c);
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_direct(float* __restrict__ e, const int* __restrict__ col, long long es, int items) {
    const int tid = threadIdx.x;
    const int G = gridDim.x, g = blockIdx.x;
    const int i0 = (int)((long long)items * g / G), i1 = (int)((long long)items * (g + 1) / G);
    for (int it = i0; it < i1; it += 2) {
        const size_t s0 = (size_t)it * 256 + tid, s1 = (size_t)min(it + 1, i1 - 1) * 256 + tid;
        float v0[6], v1[6];
        const int c0 = col[s0], c1 = col[s1];
#pragma unroll
        for (int f = 0; f < 6; ++f) v0[f] = NT ? __builtin_nontemporal_load(e + (size_t)f * es + s0) : e[(size_t)f * es + s0];
#pragma unroll
        for (int f = 0; f < 6; ++f) v1[f] = NT ? __builtin_nontemporal_load(e + (size_t)f * es + s1) : e[(size_t)f * es + s1];
#pragma unroll
        for (int f = 0; f < 6; ++f) e[(size_t)f * es + s0] = work(v0[f], c0);
        if (it + 1 < i1) {
#pragma unroll
            for (int f = 0; f < 6; ++f) e[(size_t)f * es + s1] = work(v1[f], c1);
        }
    }
}

__global__ __launch_bounds__(256) void k_wgitem(float* __restrict__ e, const int* __restrict__ col, long long es, int items) {
    const size_t s0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v0[6];
    const int c0 = col[s0];
#pragma unroll
    for (int f = 0; f < 6; ++f) v0[f] = e[(size_t)f * es + s0];
#pragma unroll
    for (int f = 0; f < 6; ++f) e[(size_t)f * es + s0] = work(v0[f], c0);
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 16384;   // x 256 slots; 16384 = 64 x dense256
    const long long es = (long long)items * 256;
    float* e;
    int* col;
    hipMalloc(&e, (size_t)es * 6 * 4);
    hipMalloc(&col, (size_t)es * 4);
    std::vector<float> he((size_t)es * 6);
    std::vector<int> hc((size_t)es);
    unsigned s = 12345;
    for (auto& v : he) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.f / 16777216.f); }
    for (auto& v : hc) { s = s * 1664525u + 1013904223u; v = (int)(s >> 20); }
    hipMemcpy(col, hc.data(), (size_t)es * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const double bytes = (double)es * (6 * 4 * 2 + 4);
    auto check = [&](const char* name) {
        std::vector<float> out((size_t)es * 6);
        hipMemcpy(out.data(), e, (size_t)es * 6 * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int f = 0; f < 6; ++f)
            for (long long k = 0; k < es; ++k) {
                const float want = fmaf(he[(size_t)f * es + k], 1.5f, (float)(hc[k] & 7));
                if (out[(size_t)f * es + k] != want) ++bad;
            }
        printf("  %-28s check: %zu mismatches of %lld\n", name, bad, es * 6);
    };
    auto run = [&](const char* name, auto launch, bool verify) {
        if (verify) {
            hipMemcpy(e, he.data(), (size_t)es * 6 * 4, hipMemcpyHostToDevice);
            launch();
            hipDeviceSynchronize();
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) { printf("  %s: %s\n", name, hipGetErrorString(err)); return; }
            check(name);
        }
        hipMemset(e, 0, (size_t)es * 6 * 4);   // zeros stay finite under repeated *1.5 + c
        for (int i = 0; i < 3; ++i) launch();
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-30s %8.2f us best, %8.2f us mean   %.2f TB/s\n", name, best * 100.f, sum * 20.f, bytes / (best * 1e-4) / 1e12);
        fflush(stdout);
    };
#define RING(R, NT, WPC)                                                                                           \
    do {                                                                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring<R, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            R * SLOT * 4);                                                                         \
        run("ring R=" #R " nt=" #NT " wg/cu=" #WPC, [&] { hipLaunchKernelGGL((k_ring<R, NT>), dim3(256 * WPC), dim3(320), R * SLOT * 4, 0, e, col, es, items); }, true); \
    } while (0)
    printf("items %d (%.1f MB read + %.1f MB written per launch)\n", items, es * 28 / 1e6, es * 24 / 1e6);
    run("wgitem", [&] { hipLaunchKernelGGL(k_wgitem, dim3(items), dim3(256), 0, 0, e, col, es, items); }, true);
    run("direct wg/cu=4", [&] { hipLaunchKernelGGL(k_direct<false>, dim3(1024), dim3(256), 0, 0, e, col, es, items); }, true);
    run("direct wg/cu=8", [&] { hipLaunchKernelGGL(k_direct<false>, dim3(2048), dim3(256), 0, 0, e, col, es, items); }, false);
    run("direct nt wg/cu=4", [&] { hipLaunchKernelGGL(k_direct<true>, dim3(1024), dim3(256), 0, 0, e, col, es, items); }, false);
    RING(4, false, 2);
    RING(4, false, 4);
    RING(6, false, 2);
    RING(6, false, 3);
    RING(8, false, 2);
    RING(6, true, 3);
    RING(10, false, 1);
    return 0;
}
