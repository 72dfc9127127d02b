#pragma once
// Shared by every kernel file: vector typedefs, error / profiling helpers, the diagnostic stamp macro.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "internal.h"

namespace gnncca {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

thread_local int g_last_hip_error = 0;

// Diagnostic build only (-DGNNCCA_STAMPS, tools/stamps.py): s_memtime stamps of every wave at named points, written to
// a buffer of their own that no kernel reads.  The product build compiles these to nothing.
#ifdef GNNCCA_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define GNNCCA_STAMP(kslot, id)                                                                              \
    do {                                                                                                     \
        if (g_stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) {                                      \
            g_stamps[((((size_t)(kslot)) * 4096 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + (id)] =       \
                __builtin_amdgcn_s_memtime();                                                                \
        }                                                                                                    \
    } while (0)
#else
#define GNNCCA_STAMP(kslot, id) \
    do {                        \
    } while (0)
#endif

// Diagnostic build only: per-wave TOTALS of the cycles spent in up to 8 phases of a loop (s_memtime deltas accumulated in registers,
// written once at the end: g_stamps[slot][block < 4096][wave & 3][0..7]).  Nothing in the product build.
#ifdef GNNCCA_STAMPS
#define PHASE_T_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_amdgcn_s_memtime(), pt_real0 = __builtin_amdgcn_s_memrealtime()
#define PHASE_T(i)                                                      \
    do {                                                                \
        const unsigned long long pt_now = __builtin_amdgcn_s_memtime(); \
        pt_acc[i] += pt_now - pt_last;                                  \
        pt_last = pt_now;                                               \
    } while (0)
#define PHASE_T_FLUSH(slot)                                                                                                   \
    do {                                                                                                                      \
 pt_acc[7] = __builtin_amdgcn_s_memrealtime() - pt_real0; /* 100 MHz */                                              \
        if (g_stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 2048 && (threadIdx.x >> 6) < 8)   /* waves 4-7: block + 2048 */  \
            for (int q = 0; q < 8; ++q)                                                                                       \
                g_stamps[((((size_t)(slot)) * 4096 + blockIdx.x + 2048 * (threadIdx.x >> 8)) * 4 + ((threadIdx.x >> 6) & 3)) * 16 + q] = pt_acc[q]; \
    } while (0)
#else
#define PHASE_T_DECL do { } while (0)
#define PHASE_T(i) do { } while (0)
#define PHASE_T_FLUSH(slot) do { } while (0)
#endif

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_last_hip_error = (int)_e;                \
            return GNNCCA_ERR_HIP;                     \
        }                                              \
    } while (0)

// Per-kernel timing for bench.py / rocprof cross-checks (diagnostic entry point only).  The events are ATTACHED TO
// THE DISPATCH (hipExtLaunchKernelGGL start / stop events), so a slot's elapsed time is the kernel's own execution
// time, the quantity rocprofv3 --kernel-trace reports -- not launch-to-launch time with the cost of an event packet
// in it (which is 2-3 us, more than half of a 5 us launch).
struct Profiler {
    gnncca_profile* out;
    hipEvent_t start[GNNCCA_PROFILE_MAX], stop[GNNCCA_PROFILE_MAX];
    int n;
};
static thread_local Profiler* t_prof = nullptr;  // non-null only inside gnncca_mpn_forward_profiled

static int prof_begin(Profiler* p) {
    p->n = 0;
    for (int i = 0; i < GNNCCA_PROFILE_MAX; ++i) {
        HIP_TRY(hipEventCreate(&p->start[i]));
        HIP_TRY(hipEventCreate(&p->stop[i]));
    }
    t_prof = p;
    return GNNCCA_OK;
}

// the launch that precedes this call used slot n (GNNCCA_LAUNCH): name it and move on
static int prof_mark(Profiler* p, int kind) {
    if (!p || p->n >= GNNCCA_PROFILE_MAX) return GNNCCA_OK;
    p->out->kind[p->n] = kind;
    p->n++;
    return GNNCCA_OK;
}

static int prof_end(Profiler* p, hipStream_t st) {
    t_prof = nullptr;
    HIP_TRY(hipStreamSynchronize(st));
    p->out->count = p->n;
    for (int i = 0; i < p->n; ++i) HIP_TRY(hipEventElapsedTime(&p->out->ms[i], p->start[i], p->stop[i]));
    for (int i = 0; i < GNNCCA_PROFILE_MAX; ++i) {
        HIP_TRY(hipEventDestroy(p->start[i]));
        HIP_TRY(hipEventDestroy(p->stop[i]));
    }
    return GNNCCA_OK;
}

#define PROF_MARK(kind)                                  \
    do {                                                 \
        int _s = prof_mark(prof, (kind));                \
        if (_s != GNNCCA_OK) return _s;                  \
    } while (0)

// One scalar load per 64-byte line of the kernel-argument segment, all in flight behind the kernel's FIRST scalar wait.  The compiler fetches
// kernel arguments lazily, cluster by cluster, each right before its first use; the scalar cache is cold at a launch's start, so every cluster
// on a new line is an L2 round trip of its own -- four or five of them, one after the other, between a wave's start and its first vector
// load in the latency-bound step kernels.  Touched up front the lines arrive together and the later fetches hit the scalar cache.
// (A value, not a statement: the caller hands it to the asm that pins its first round of scalar loads -- an `asm volatile` AHEAD of those
// loads would make the compiler treat the memory behind them as clobbered and turn them into vector loads.)
__device__ __forceinline__ int touch_kernargs(unsigned bytes) {
    typedef const int __attribute__((address_space(4))) cint;
    cint* ka = (cint*)__builtin_amdgcn_kernarg_segment_ptr();
    int t = 0;
    if (bytes > 64) t |= ka[16];
    if (bytes > 128) t |= ka[32];
    if (bytes > 192) t |= ka[48];
    if (bytes > 256) t |= ka[64];
    if (bytes > 320) t |= ka[80];
    if (bytes > 384) t |= ka[96];
    return t;
}

// Kernel launch of the forward path: plain, or with the profiler's events attached to this very dispatch.
#define GNNCCA_LAUNCH(kernel, grid, block, lds, st, ...)                                                              \
    do {                                                                                                              \
        if (t_prof != nullptr && t_prof->n < GNNCCA_PROFILE_MAX)                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t_prof->start[t_prof->n], t_prof->stop[t_prof->n], 0, \
                                  __VA_ARGS__);                                                                       \
        else                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                            \
    } while (0)


static inline dim3 grid1(size_t n, int b) { return dim3((unsigned)((n + b - 1) / b)); }

// ---- train-mode Dropout (models/mlp.py:20-21: nn.Dropout after the ReLU of every MLP layer wider than 1) ---------------------------
// Masks are never stored: element `idx` of the activation tensor identified by `stream` is kept iff a counter-based hash of
// (seed, stream, idx) says so, and forward and backward evaluate the same hash.  idx = row * width + column with row = the node
// id or the CALLER's edge id.  oracle/mpn_oracle.py:dropout_keep is the numpy twin (tests, goldens).
enum : unsigned {
    kDropEncNode1 = 1, kDropEncNode2 = 2, kDropEncEdge = 3,
    kDropEdgeStep = 16,   // + step (1-based)
    kDropNodeStep = 48,   // + step
    kDropCls = 80,        // + index of the classified step (0-based)
};
struct DropCfg {          // device-side view of gnncca_dropout; p == 0 everywhere when dropout is off
    float p_enc, p_edge, p_node, p_cls;
    const unsigned long long* seed;   // device word, read by the kernels (a captured training step replays with fresh masks)
};
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, unsigned stream, unsigned long long idx) {
    unsigned long long x = idx * 0x9E3779B97F4A7C15ull + (seed ^ ((unsigned long long)stream * 0xD1B54A32D192ED03ull));
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (unsigned)x;
}
// 0 (dropped) or 1 / (1 - p) (kept): uniform u = (hash >> 8) * 2^-24 in [0, 1), kept iff u >= p
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned stream, unsigned long long idx, float p) {
    const float u = (float)(drop_hash(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace gnncca
