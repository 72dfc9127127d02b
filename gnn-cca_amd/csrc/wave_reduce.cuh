#pragma once
// Wave-level reductions shared by mpn_forward.hip and graph_build.hip.
#include <hip/hip_runtime.h>

namespace gnncca {

// 16 per-lane partial sums v[idx]  ->  the full 64-lane sum of v[lane >> 2] in every lane (lanes 4 idx .. 4 idx + 3
// hold the sum of idx).  A transposing butterfly: each exchange halves the number of values a lane still carries --
// 8 v_permlane32_swap, 4 v_permlane16_swap, then 2 + 1 + 2 shuffles: 17 cross-lane operations where 16 independent
// butterflies would need 96.
__device__ __forceinline__ float transpose_reduce16(const float (&v)[16]) {
    float w[8], x[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // lanes 0-31 keep idx k, lanes 32-63 keep idx k + 8
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
        w[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // odd rows of 16 lanes keep idx + 4
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[k]), __float_as_uint(w[k + 4]), false, false);
        x[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const int lane = threadIdx.x & 63;
    const bool b3 = lane & 8, b2 = lane & 4;
    const float y0 = (b3 ? x[2] : x[0]) + __shfl_xor(b3 ? x[0] : x[2], 8);
    const float y1 = (b3 ? x[3] : x[1]) + __shfl_xor(b3 ? x[1] : x[3], 8);
    float z = (b2 ? y1 : y0) + __shfl_xor(b2 ? y0 : y1, 4);
    z += __shfl_xor(z, 2);
    z += __shfl_xor(z, 1);
    return z;
}

// after transpose_reduce16 the full sum of idx sits in lanes 4 * idx .. 4 * idx + 3
__host__ __device__ constexpr int lane_of_sum(int idx) { return 4 * idx; }

}  // namespace gnncca
