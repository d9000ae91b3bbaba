// "fh2": an fp32 matrix held as TWO fp16 planes, x ~ (h0 + h1) / s with h0 = rn_f16(s x), h1 = rn_f16(s x - h0) and s a power of
// two chosen per matrix (1 for activations, 2^k putting max|w| into [2^12, 2^13) for weights).  h0 + h1 carries 22 significant bits of
// s x (fp32 has 24) whenever |s x| >= 2^-3 -- below that h1 is an fp16 subnormal and the ABSOLUTE error is bounded by 2^-25 / s.
// An fp32 GEMM is then evaluated on the fp16 matrix cores as
//     a b ~ h0 g0 + (h0 g1 + h1 g0)          [dropped: h1 g1 <= 2^-22 |a b|]
// THREE v_mfma_f32_16x16x32_f16 passes per k-step, every fp16 x fp16 product exact in the fp32 accumulator.  The operand error
// (2^-22 relative, random sign) is averaged over the K terms of a dot product and sits well below the fp32 accumulation error that
// the reference's own fp32 GEMM has: measured against float64 the result is as accurate as the exact-fp32 MFMA kernel's
// (tests/test_gpu_fh2.py: max |err| / sum|a||b| within 1.2x of a3r_linear's), at half the matrix passes and two thirds of the
// operand bytes of the exact three-plane bf16 form (bf3.h), which stays available (A3R_GEMM=bf3).
// RANGE.  fp16 ends at 65504 and h1 turns subnormal where |s x| < 2^-3, so an fh2 tensor is only fp32-grade while its largest |s x| stays
// inside [2^-2, 2^15].  Weights get their s once from max|w|.  Activations start at s = 1; every kernel that WRITES an fh2 tensor takes
// the power of two to store it with and a device word that receives max |s x| (atomic max on the bit pattern), every kernel that READS
// one takes the s it was stored with and divides it out exactly.  The model plan (model.hip) keeps one s per producing site and
// a3r_model_range_check() moves the sites whose maximum left the band -- an out-of-range checkpoint costs one repeated forward, not a
// wrong or refused result.
// Memory layout of a [R, K] matrix (K % 32 == 0): [R][K/8][2][8] fp16 -- per row and group of 8 consecutive k the two planes'
// 16-byte pieces are adjacent (32 bytes); row pitch 4 K bytes; the 32 k of one GEMM stage are exactly one 128-byte line.
#pragma once
#include "common.h"

namespace a3r {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float fh2_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t fh2_u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t fh2_row_bytes(int K) { return (size_t)K * 4; }

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {       // v_cvt_pkrtz is round-to-zero: use the rn conversions
    fh2_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));
}
__device__ __forceinline__ float f16lo(uint32_t p) { return (float)__builtin_bit_cast(f16x2v, p)[0]; }
__device__ __forceinline__ float f16hi(uint32_t p) { return (float)__builtin_bit_cast(f16x2v, p)[1]; }

// two (already scaled) floats -> two packed fp16 pairs (low half = a, high half = b).  The second plane rn_f16(x - h0) is ONE
// v_fma_mix per element (f32 x, f16 h0 read straight from the packed pair, f16 result into its half of the destination) instead of
// convert-back, subtract and convert again: x - h0 is exact in fp32, so both forms round the same number once -- identical bits, half
// the instructions of the split (the attention kernel splits 16 probabilities per lane and 32-key block beside its MFMAs)
__device__ __forceinline__ void fh2_split2(float a, float b, uint32_t& p0, uint32_t& p1) {
    p0 = pk_f16(a, b);
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(p0));
    p1 = r;
}

// eight consecutive k (k0 % 8 == 0) of one fh2 row: 32 contiguous bytes
__device__ __forceinline__ void fh2_store8(char* row, int k0, f32x4 lo, f32x4 hi) {
    uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
    fh2_split2(lo.x, lo.y, a0, a1);
    fh2_split2(lo.z, lo.w, b0, b1);
    fh2_split2(hi.x, hi.y, c0, c1);
    fh2_split2(hi.z, hi.w, d0, d1);
    char* d = row + (k0 >> 3) * 32;
    *reinterpret_cast<fh2_u32x4*>(d) = fh2_u32x4{a0, b0, c0, d0};
    *reinterpret_cast<fh2_u32x4*>(d + 16) = fh2_u32x4{a1, b1, c1, d1};
}

// four consecutive k (k0 % 4 == 0): half of each plane's 16-byte unit
__device__ __forceinline__ void fh2_store4(char* row, int k0, f32x4 v) {
    uint32_t a0, a1, b0, b1;
    fh2_split2(v.x, v.y, a0, a1);
    fh2_split2(v.z, v.w, b0, b1);
    char* d = row + (k0 >> 3) * 32 + ((k0 >> 2) & 1) * 8;
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2_*>(d) = u32x2_{a0, b0};
    *reinterpret_cast<u32x2_*>(d + 16) = u32x2_{a1, b1};
}

// ---- range statistics: running max |v| (v_max3_f32 with |.| source modifiers: half an instruction per element)
__device__ __forceinline__ float fh2_amax2(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }
__device__ __forceinline__ float fh2_amax4(float m, f32x4 v) { return fh2_amax2(fh2_amax2(m, v.x, v.y), v.z, v.w); }
// Publishing the maximum: ONE fire-and-forget atomic per workgroup (per-wave butterfly -> LDS -> thread 0).  No read of the slot
// first: a load whose result gates the atomic keeps every wave alive for an L2 round trip at its very end -- with one wave per
// LayerNorm row that doubled the kernel's duration (round 3, measured).  Kernels keep their workgroup count in the low thousands
// (persistent loops) so the atomics of a launch do not queue up on one address.  A non-negative float compares like its bit pattern;
// Inf sorts above every finite value, NaN above Inf.  Every thread of the workgroup must call it.  s_red: blockDim / 64 words of LDS.
__device__ __forceinline__ void fh2_publish_block(unsigned* slot, unsigned b, unsigned* s_red) {
    if (!slot) return;                                   // uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = s_red[0];
        for (unsigned w = 1; w < (blockDim.x >> 6); w++) m = max(m, s_red[w]);
        if (m) atomicMax(slot, m);
    }
}
__device__ __forceinline__ void fh2_publish_block(unsigned* slot, float amax, unsigned* s_red) {
    fh2_publish_block(slot, __float_as_uint(amax), s_red);
}

// NaN-aware statistics for the HBM-bound producers that read MODEL INPUTS (split passes: images, point maps, weights): the maximum of
// the bit patterns with the sign cleared, where NaN sorts above Inf -- v_max_f32 ignores a NaN operand, and a NaN that enters the
// network can end as a finite number (fmaxf(NaN, 0) = 0 in a ReLU), so it must be caught where it enters.  Two VALU ops per element.
__device__ __forceinline__ unsigned fh2_amax_bits4(unsigned m, f32x4 v) {
    const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
    return max(max(m, max(a, b)), max(c, d));
}

}  // namespace a3r
