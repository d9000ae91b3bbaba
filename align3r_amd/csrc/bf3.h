// "bf3": an fp32 matrix held as three bf16 planes, x = x0 + x1 + x2 exactly (x0 = rn_bf16(x), x1 = rn_bf16(x - x0),
// x2 = x - x0 - x1, which has at most 8 significant bits left), so that fp32 GEMMs can run on the bf16 matrix
// cores: a*b = sum_{p,q} a_p b_q with every bf16 x bf16 product exact in the fp32 accumulator (gemm_bf3.hip).
// Memory layout of a [R, K] matrix (K % 8 == 0): [R][K/8][3][8] bf16 -- per row and per group of 8 consecutive k
// the three planes' 16-byte pieces are adjacent (48 bytes), which is exactly one lane's A/B operand of
// v_mfma_f32_32x32x16_bf16 per plane.  Row pitch = 6 K bytes.
//
// WEIGHT matrices (the B operand of gemm_bf3.hip; K % 32 == 0) use the ROW-PAIR form of the same layout,
// [ceil(N/2)][K/32][2 rows][4 k-groups][3][8] bf16: rows 2j and 2j+1 are interleaved per 32-deep k block, so the 32 k of a row
// pair that one GEMM stage needs are 384 contiguous bytes = three whole 128-byte cache lines.  In the plain form a stage
// takes 192 bytes = 1.5 lines of every row, i.e. it drags 2 lines through the L2 -> LDS path for 1.5 lines of payload;
// measured on MI355X (tools/gemm_bf3_lab.hip, "RB2"): +2-3.5 % GEMM throughput from the weights alone, +4.5-5.5 % if the A
// operand is stored the same way (not done: activations are produced and sliced by row in many kernels).
//   byte offset of (row n, k) = (n / 2) * 12 K + (k / 32) * 384 + (n % 2) * 192 + ((k / 8) % 4) * 48 [+ plane * 16 + (k % 8) * 2]
#pragma once
#include "common.h"

namespace a3r {

__host__ __device__ inline size_t bf3_w_row_offset(int n, int K) { return (size_t)(n >> 1) * ((size_t)12 * K) + (size_t)(n & 1) * 192; }
constexpr int BF3_W_KBLOCK_BYTES = 384;      // one 32-deep k block of a weight row pair
// The same row-pair form serves ACTIVATION matrices that are only ever read as the A operand of a GEMM (`pair` != 0 below): the
// outputs of LayerNorm, of the attention kernel and of the fc1 + GELU epilogue in the transformer blocks.  pair == 0: plain rows.
__host__ __device__ inline size_t bf3_row_offset(long r, int K, int pair) {
    return pair ? (size_t)(r >> 1) * ((size_t)12 * K) + (size_t)(r & 1) * 192 : (size_t)r * ((size_t)6 * K);
}
__host__ __device__ inline int bf3_k_offset(int k, int pair) {        // of the 48-byte group holding k (k % 8 ignored)
    return pair ? (k >> 5) * 384 + ((k >> 3) & 3) * 48 : (k >> 3) * 48;
}

int bf3_products();        // 6 | 3 | 1: the process-wide product set of the bf3 kernels (gemm_bf3.hip)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even)
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}

// two floats -> three packed bf16 pairs (low half = a, high half = b)
__device__ __forceinline__ void bf3_split2(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
}

// four consecutive k (k0 % 4 == 0) of one bf3 row
__device__ __forceinline__ void bf3_store4(char* row, int k0, f32x4 v, int pair = 0) {
    char* d = row + bf3_k_offset(k0, pair) + ((k0 >> 2) & 1) * 8;
    uint32_t a0, a1, a2, b0, b1, b2;
    bf3_split2(v.x, v.y, a0, a1, a2);
    bf3_split2(v.z, v.w, b0, b1, b2);
    *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(d + 16) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(d + 32) = u32x2{a2, b2};
}

// eight consecutive k (k0 % 8 == 0) of one bf3 row: 48 contiguous bytes
__device__ __forceinline__ void bf3_store8(char* row, int k0, f32x4 lo, f32x4 hi, int pair = 0) {
    char* d = row + bf3_k_offset(k0, pair);
    uint32_t a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
    bf3_split2(lo.x, lo.y, a0, a1, a2);
    bf3_split2(lo.z, lo.w, b0, b1, b2);
    bf3_split2(hi.x, hi.y, c0, c1, c2);
    bf3_split2(hi.z, hi.w, d0, d1, d2);
    *reinterpret_cast<u32x4*>(d) = u32x4{a0, b0, c0, d0};
    *reinterpret_cast<u32x4*>(d + 16) = u32x4{a1, b1, c1, d1};
    *reinterpret_cast<u32x4*>(d + 32) = u32x4{a2, b2, c2, d2};
}

}  // namespace a3r
