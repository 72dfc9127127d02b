#pragma once
// Part of the single translation unit mpn_forward.hip.
namespace gnncca {

// ------------------------------------------------------------------------------------------------------------
// The node message  m[edge][ch] = Q[row][ch] + sum_k W_ne[ch][k] * e'[edge][k]   (models/mpn.py:97-98, before the
// ReLU) on the bf16 matrix pipe at fp32 accuracy.
//
// Why not v_mfma_f32_32x32x2_f32 (rounds 1-2): measured on this chip (tools/ubench_mfma_valu.hip, round 3) an f32-input
// MFMA and the VALU of the same SIMD do NOT overlap, not even across waves -- 8 MFMAs + 32 v_fma take 290 ns where the
// MFMAs alone take 225 and the FMAs alone 73 -- while a bf16-input MFMA runs beside the VALU nearly for free (123 ns
// against 117 + 73).  The step kernel with every HBM stream switched off still took 80 % of its time: it is bound by
// SIMD issue, of which the f32 MFMAs were 40 %.
//
// Split form: every fp32 operand is the exact sum of three bf16 pieces (round-to-nearest-even, twice on the residual:
// x = a0 + a1 + a2 up to 2^-24 |x|), bf16 x bf16 products are exact in fp32, and the MFMA accumulates in fp32.  Six of
// the nine piece products are kept (a0b0, a0b1, a1b0, a0b2, a2b0, a1b1); the dropped ones are <= 2^-23 of the product.
// K = 16 per v_mfma_f32_32x32x16_bf16, so the 6 x 6 = 36 slots + bias + tail mask fit THREE MFMAs per 32-edge tile
// (lane halves = k halves of the instruction), and the pairing below needs only TWO distinct A operands per tile:
//     M1:  A = (a0 | a1)   k 0..5  a0 x b0 | k 6,7  1 x (q0, q1)     || k 8..13 a1 x b0 | k 14  1 x q2 | k 15  dead-edge x -3e38
//     M2:  A = (a0 | a1)   k 0..5  a0 x b1 | k 6,7  1 x 0            || k 8..13 a1 x b1 | k 14, 15  . x 0           (same A registers)
//     M3:  A = (a0 | a2)   k 0..5  a0 x b2 | 0                       || k 8..13 a2 x b0 | 0
// q0 + q1 + q2 = Q[row][ch] (the bias rides in the product, the accumulator starts from the inline constant 0); an edge
// beyond the segment gets -3e38 before the ReLU, i.e. 0 after it.  96 matrix-pipe cycles per tile that overlap with the
// VALU instead of 192-256 that do not.  Operand construction is VALU work: 27 instructions to split a chunk's six
// features, 6 v_permlane32_swap (+ 3 copies) to put "lane = edge" pieces into the A layout of both tiles.
// The general kernel (step_general.cuh) and the fast kernel (step_pipe.cuh) call the same functions: same bits.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned msg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
// (x0, x1) -> three packed bf16 pairs with p0 + p1 + p2 = x (to 2^-24)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(p1 << 16), s1 = r1 - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = cvt_pk_bf16(s0, s1);
}

struct MsgB {            // B operands (lane: channel = lane & 31, k half = lane >> 5), three packed pairs per MFMA
    unsigned b0[3];      // b0 pieces of W_ne[ch][2j, 2j+1] in both halves                           (M1)
    unsigned b1[3];      // b1 pieces in both halves                                                 (M2)
    unsigned b2[3];      // lanes < 32: b2 pieces; lanes >= 32: b0 pieces                            (M3)
    unsigned bq;         // lanes < 32: (q0, q1) of Q[node][ch]; lanes >= 32: (q2, -3e38)            (M1, fourth register)
};
// W_ne pieces as packed at weight-pack time (BlobHeader::wne_bf16: [9][64] dwords in exactly this lane layout)
__device__ __forceinline__ void msg_b_weights(const float* __restrict__ wne_bf16, int lane, MsgB& B) {
    const unsigned* __restrict__ w = reinterpret_cast<const unsigned*>(wne_bf16);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        B.b0[j] = w[j * 64 + lane];
        B.b1[j] = w[(3 + j) * 64 + lane];
        B.b2[j] = w[(6 + j) * 64 + lane];
    }
}
__device__ __forceinline__ void msg_b_bias(float q, int lane, MsgB& B) {   // q = Q[node][lane & 31]
    const unsigned t0 = cvt_pk_bf16(q, 0.f);
    const float r = q - __uint_as_float(t0 << 16);
    const unsigned t1 = cvt_pk_bf16(r, 0.f);
    const float s = r - __uint_as_float(t1 << 16);
    const unsigned t2 = cvt_pk_bf16(s, -3.0e38f);                    // (q2, -3e38)
    B.bq = (lane >> 5) ? t2 : ((t0 & 0xFFFFu) | (t1 << 16));         // (q0, q1)
}

struct MsgA {            // A operands of one 64-edge chunk: [tile] (tile 0 = edges 0..31 of the chunk, tile 1 = 32..63)
    msg_u32x4 a01[2];    // k 0..7: a0 pieces + (1, 1); k 8..15: a1 pieces + (1, dead)      (M1 and M2)
    msg_u32x4 a02[2];    // k 0..7: a0 pieces;          k 8..15: a2 pieces                  (M3)
};
// en = e' of this lane's edge; `base` = index of the chunk's first edge, seg_t = end of the segment (tail mask)
__device__ __forceinline__ void msg_a_operands(const float (&en)[kEF], int base, int seg_t, int lane, MsgA& A) {
    const int half = lane >> 5, km = base + (lane & 31);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned p0, p1, p2;
        split3_pair(en[2 * j], en[2 * j + 1], p0, p1, p2);
        // v_permlane32_swap(x, y): r[0] = [x.lanes 0-31 | y.lanes 0-31], r[1] = [x.lanes 32-63 | y.lanes 32-63]
        auto r = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
        A.a01[0][j] = r[0], A.a01[1][j] = r[1];
        r = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);
        A.a02[0][j] = r[0], A.a02[1][j] = r[1];
    }
    // fourth register: lanes < 32 (k 6, 7) = (1, 1) for the bias pieces q0, q1; lanes >= 32 (k 14, 15) = (1, dead ? 1 : 0)
    A.a01[0][3] = (half && km < seg_t) ? 0x00003F80u : 0x3F803F80u;
    A.a01[1][3] = (half && km + 32 < seg_t) ? 0x00003F80u : 0x3F803F80u;
    A.a02[0][3] = A.a02[1][3] = 0u;
}

__device__ __forceinline__ f32x16 msg_mfma(msg_u32x4 a, unsigned b0, unsigned b1, unsigned b2, unsigned b3, f32x16 c) {
    const msg_u32x4 b = {b0, b1, b2, b3};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 msg_m1(const MsgA& A, const MsgB& B, int t) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return msg_mfma(A.a01[t], B.b0[0], B.b0[1], B.b0[2], B.bq, z);
}
__device__ __forceinline__ f32x16 msg_m2(const MsgA& A, const MsgB& B, int t, f32x16 d) {
    return msg_mfma(A.a01[t], B.b1[0], B.b1[1], B.b1[2], 0u, d);
}
__device__ __forceinline__ f32x16 msg_m3(const MsgA& A, const MsgB& B, int t, f32x16 d) {
    return msg_mfma(A.a02[t], B.b2[0], B.b2[1], B.b2[2], 0u, d);
}
// the whole message of tile t (pre-activation), for callers that do not interleave
__device__ __forceinline__ f32x16 msg_tile(const MsgA& A, const MsgB& B, int t) {
    return msg_m3(A, B, t, msg_m2(A, B, t, msg_m1(A, B, t)));
}

}  // namespace gnncca
