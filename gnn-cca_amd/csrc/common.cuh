#pragma once
// Shared by every kernel file: vector typedefs, error / profiling helpers, the diagnostic stamp macro.
#include <hip/hip_runtime.h>

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

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_last_hip_error = (int)_e;                \
            return GNNCCA_ERR_HIP;                     \
        }                                              \
    } while (0)

// Per-kernel timing for bench.py / rocprof cross-checks: one hipEvent after every launch (diagnostic entry point only).
struct Profiler {
    gnncca_profile* out;
    hipEvent_t ev[GNNCCA_PROFILE_MAX + 1];
    int n;
};

static int prof_begin(Profiler* p, hipStream_t st) {
    if (!p) return GNNCCA_OK;
    p->n = 0;
    for (int i = 0; i <= GNNCCA_PROFILE_MAX; ++i) HIP_TRY(hipEventCreate(&p->ev[i]));
    HIP_TRY(hipEventRecord(p->ev[0], st));
    return GNNCCA_OK;
}

static int prof_mark(Profiler* p, int kind, hipStream_t st) {
    if (!p || p->n >= GNNCCA_PROFILE_MAX) return GNNCCA_OK;
    p->out->kind[p->n] = kind;
    p->n++;
    HIP_TRY(hipEventRecord(p->ev[p->n], st));
    return GNNCCA_OK;
}

static int prof_end(Profiler* p, hipStream_t st) {
    if (!p) return GNNCCA_OK;
    HIP_TRY(hipStreamSynchronize(st));
    p->out->count = p->n;
    for (int i = 0; i < p->n; ++i) HIP_TRY(hipEventElapsedTime(&p->out->ms[i], p->ev[i], p->ev[i + 1]));
    for (int i = 0; i <= GNNCCA_PROFILE_MAX; ++i) HIP_TRY(hipEventDestroy(p->ev[i]));
    return GNNCCA_OK;
}

#define PROF_MARK(kind)                                  \
    do {                                                 \
        int _s = prof_mark(prof, (kind), st);            \
        if (_s != GNNCCA_OK) return _s;                  \
    } while (0)


static inline dim3 grid1(size_t n, int b) { return dim3((unsigned)((n + b - 1) / b)); }

}  // namespace gnncca
