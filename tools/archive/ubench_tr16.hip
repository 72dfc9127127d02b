// check of the transposing butterfly used by graph_build.hip: prints, per lane, which idx's sum it ends up holding
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float transpose_reduce16(float (&v)[16], float* dbg) {
    float w[8], x[4];
    for (int k = 0; k < 8; ++k) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
        w[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    for (int k = 0; k < 4; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[k]), __float_as_uint(w[k + 4]), false, false);
        x[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const int lane = threadIdx.x & 63;
    dbg[lane] = w[0]; dbg[64 + lane] = x[0];
    const bool b3 = lane & 8, b2 = lane & 4;
    const float y0 = (b3 ? x[2] : x[0]) + __shfl_xor(b3 ? x[0] : x[2], 8);
    const float y1 = (b3 ? x[3] : x[1]) + __shfl_xor(b3 ? x[1] : x[3], 8);
    float z = (b2 ? y1 : y0) + __shfl_xor(b2 ? y0 : y1, 4);
    z += __shfl_xor(z, 2);
    z += __shfl_xor(z, 1);
    return z;
}
__global__ void k(float* out, float* dbg) {
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = q * 1000.f + 1.f;   // sum over 64 lanes = 64000 q + 64
    out[threadIdx.x] = transpose_reduce16(v, dbg);
}
int main() {
    float *d, *g, h[64], hd[128];
    hipMalloc(&d, 256); hipMalloc(&g, 512);
    k<<<1, 64>>>(d, g);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, g, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: tot %.0f -> idx %.2f | w0 %.0f x0 %.0f\n", l, h[l], (h[l] - 64) / 64000, hd[l], hd[64 + l]);
    return 0;
}
