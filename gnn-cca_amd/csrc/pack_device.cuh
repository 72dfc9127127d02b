#pragma once
// Part of the single translation unit mpn_forward.hip.
// Device-side weight packing: interprets the PackProgram of pack.cpp on the GPU, so that a training step (the
// optimizer has just changed every parameter) or a load_state_dict() never moves the parameters through the host.
// Same arithmetic as the host packer, operation for operation: BatchNorm fold in double with separately rounded
// multiply / add (no FMA contraction), bf16 pieces by the same integer round-to-nearest-even.
namespace gnncca {

constexpr int kMaxPackParams = 96;
struct PackPtrs {
    const float* p[kMaxPackParams];
};

__device__ __forceinline__ unsigned short pack_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void pack_device_kernel(const PackProgram* __restrict__ prog, const PackPtrs ptrs,
                                                          float* __restrict__ blob) {
    const int n_segs = prog->n_segs;
    if ((int)blockIdx.y == n_segs) {  // the header
        const unsigned* h = reinterpret_cast<const unsigned*>(&prog->header);
        for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < sizeof(BlobHeader) / 4; t += gridDim.x * 256)
            reinterpret_cast<unsigned*>(blob)[t] = h[t];
        return;
    }
    const PackSeg g = prog->segs[blockIdx.y];
    const float* __restrict__ src = ptrs.p[g.param];
    const float *gamma = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
    if (g.bn >= 0) gamma = ptrs.p[g.bn], beta = ptrs.p[g.bn + 1], mean = ptrs.p[g.bn + 2], var = ptrs.p[g.bn + 3];
    const long long total = (long long)g.rows * g.cols;
    int f16_bad = 0;   // kind 4: an element of this block does not fit fp16
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int r = (int)(t / g.cols), c = (int)(t - (long long)r * g.cols);
        float v = src[(size_t)g.src_off + (size_t)r * g.srs + (size_t)c * g.scs];
        if (gamma) {  // BatchNorm1d eval, eps = 1e-5 (models/mlp.py:15)
            const int u = g.unit0 + r;
            const double s = __ddiv_rn((double)gamma[u], __dsqrt_rn(__dadd_rn((double)var[u], 1e-5)));
            if (g.kind == 1)
                v = (float)__dadd_rn(__dmul_rn(__dsub_rn((double)v, (double)mean[u]), s), (double)beta[u]);
            else
                v = (float)__dmul_rn((double)v, s);
        }
        size_t k = (size_t)r * g.drs + (size_t)c * g.dcs;
        if (g.kind == 2) k = (size_t)(c / 32) * 3 * (size_t)g.plane + (size_t)r * 32 + (size_t)(c % 32);  // [in/32][3][out][32]
        if (g.kind == 3) {   // W_ne[ch = r][k = c] as bf16 pieces in the MsgB lane layout (see gnncca_pack_weights)
            unsigned short* wb = reinterpret_cast<unsigned short*>(blob + g.dst);
            const unsigned short h0 = pack_bf16_rne(v);
            const float r1 = v - __uint_as_float((unsigned)h0 << 16);
            const unsigned short h1 = pack_bf16_rne(r1);
            const float r2 = r1 - __uint_as_float((unsigned)h1 << 16);
            const unsigned short h2 = pack_bf16_rne(r2);
            const size_t j2 = (size_t)(c / 2) * 128 + (c & 1);
            wb[j2 + 2 * r] = wb[j2 + 2 * (r + 32)] = h0;
            wb[384 + j2 + 2 * r] = wb[384 + j2 + 2 * (r + 32)] = h1;
            wb[768 + j2 + 2 * r] = h2;
            wb[768 + j2 + 2 * (r + 32)] = h0;
        } else if (g.kind == 4) {   // two fp16 pieces in the swizzled chunk image of enc_f16.cuh (pack.cpp: w2h_index); `(_Float16)` is
            // v_cvt_f16_f32: round to nearest even, subnormals kept -- pack.cpp's f16_rne bit for bit
            _Float16* w2 = reinterpret_cast<_Float16*>(blob + g.dst);
            const _Float16 h0 = (_Float16)v;
            const float rr = (v - (float)h0) * 2048.0f;
            const int kq = c % 32, ln = (r % 32) + 32 * ((kq % 16) / 8);   // (pack.cpp: w2h_index -- fragment-major inside the chunk)
            const size_t kk = (size_t)(c / 32) * (2 * 128 * 32) + (size_t)((((r / 32) * 2 + kq / 16) * 64 + ln) * 8 + (kq % 8));
            w2[kk] = h0;
            w2[kk + 128 * 32] = (_Float16)rr;
            if (!(fabsf(v) < kF16Limit)) f16_bad = 1;
        } else if (g.kind == 2) {
            unsigned short* w3 = reinterpret_cast<unsigned short*>(blob + g.dst);
            const unsigned short h0 = pack_bf16_rne(v);
            const float r1 = v - __uint_as_float((unsigned)h0 << 16);
            const unsigned short h1 = pack_bf16_rne(r1);
            const float r2 = r1 - __uint_as_float((unsigned)h1 << 16);
            w3[k] = h0;
            w3[(size_t)g.plane + k] = h1;
            w3[2 * (size_t)g.plane + k] = pack_bf16_rne(r2);
        } else {
            blob[(size_t)g.dst + k] = v;
        }
    }
    if (g.kind == 4) {   // (block-uniform) this block's overflow word: written every time, so a repack needs no reset and no atomics
        const int any = __syncthreads_or(f16_bad);
        if (threadIdx.x == 0 && blockIdx.x < kW2hBadWords) reinterpret_cast<unsigned*>(blob + g.plane)[blockIdx.x] = any ? 1u : 0u;
    }
}

}  // namespace gnncca
