// Flash-style attention for head_dim 64 on the fp16 matrix cores from two-plane fp16 operands ("fh2", fh2.h).
//
// Replaces  attn = softmax(q @ k^T * hd^-0.5); x = attn @ v   of
//   Attention.forward      croco/models/blocks.py:105-109
//   CrossAttention.forward croco/models/blocks.py:164-168
// like attention_bf3.hip, with every product evaluated as THREE exact fp16 x fp16 MFMA passes (h0 g0 + h0 g1 + h1 g0) instead of
// six bf16 ones: q, k, v arrive in fh2 form from the projection GEMM (RoPE + out_fh2 epilogue, gemm_fh2.hip), S^T = K Q^T and
// O^T = V^T P^T run on v_mfma_f32_32x32x16_f16, the online softmax is fp32 on the accumulators, and P -- in [0, 1] -- is split into
// two fp16 planes of 1024 p in registers (22 significant bits down to p = 2^-13, absolute error 2^-35 below that) and used directly
// as the B operand; the factor 1024 is divided out with the softmax normaliser.  O is written in fh2 form for the output
// projection.  Structure, staging and LDS layouts as attention_bf3.hip (K tiles by LDS-DMA into a double buffer, V tiles through
// registers into a transposed, key-permuted image, two 32-key online-softmax blocks per 64-key tile); the two-plane images are
// 53 KB per workgroup, so THREE workgroups of 4 waves share a CU.
#include "common.h"
#include "fh2.h"
#include <cmath>
#include <atomic>
#include <cstdlib>
#include <string>

namespace a3r {
int fh2_passes();      // gemm_fh2.hip: 3 (default) or 1 (a3r_fh2_set_passes)

constexpr int A4T = 256;                    // threads per workgroup (4 waves, 32 queries each)
constexpr int A4Q = 128;                    // queries per workgroup
constexpr int A4K = 64;                     // keys per tile
constexpr int A4_KUNITS = 17;               // K row: 16 units (8 d-groups x 2 planes) + 1 pad unit (an odd stride: conflict-free b128 reads)
constexpr int A4_KROW = A4_KUNITS * 16;
constexpr int A4_KSLOTS = A4K * A4_KUNITS;  // 1088 DMA slots per tile
constexpr int A4KI = (A4_KSLOTS + A4T - 1) / A4T;   // 5 per thread, the last one only on wave 0
constexpr int A4VI = (A4K * 16) / A4T;      // 4 V units per thread
constexpr int A4_KS_BYTES = A4K * A4_KROW;  // one K tile (two of them)
constexpr int A4_VROW = 9 * 16;             // V^T row: 64 keys x 2 B + 1 pad unit
constexpr int A4_VT_BYTES = 2 * 64 * A4_VROW;   // V^T tile: 2 planes x 64 d
constexpr int A4_LDS_BYTES = 2 * A4_KS_BYTES + A4_VT_BYTES;   // 53,248 B
constexpr float A4_PSHIFT = 10.f;     // P is carried as 2^10 p in its two fp16 planes (p <= 1: 1024 p keeps 21+ bits above fp16's subnormal range)

struct Attn4Args {
    const char *q, *k, *v;
    char* o;
    size_t pq, pk, pv, po;                  // row pitches in bytes (4 x leading dimension)
    int B, H, Nq, Nk;
    float scale_log2e;                      // hd^-0.5 log2(e) / (q_scale k_scale): the exp2 argument's factor on the raw scores
    float out_mul;                          // out_scale / v_scale, applied with the softmax normaliser
    unsigned* out_absmax;                   // range statistics of the stored output (fh2.h), or null
};

typedef const __attribute__((address_space(1))) void* a4_gptr;
typedef __attribute__((address_space(3))) void* a4_lptr;
typedef uint32_t a4_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(A4T, 2) void attn_fh2_kernel(Attn4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                // [2][64 rows][17 units]
    char* Vt = smem + 2 * A4_KS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, half = lane >> 5;
    // XCD-aware mapping: all query blocks of one (batch, head) land on ONE XCD
    const int nqb = (a.Nq + A4Q - 1) / A4Q;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int group = (seq / nqb) * 8 + xcd, qb = seq - (seq / nqb) * nqb;
    if (group >= a.B * a.H) return;                 // uniform per workgroup
    const int h = group % a.H, b = group / a.H;
    const int q_row = qb * A4Q + wave * 32 + qi;
    const int q_ld = q_row < a.Nq ? q_row : a.Nq - 1;

    // Q operand (B of S^T = K Q^T): lane (query, half) holds, per 16-deep d-step s and plane p, unit (2 s + half) 2 + p of its head
    f16x8 qf[4][2];
    {
        const char* qp = a.q + ((size_t)b * a.Nq + q_ld) * a.pq + h * 256;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int p = 0; p < 2; p++) qf[s][p] = *reinterpret_cast<const f16x8*>(qp + ((2 * s + half) * 2 + p) * 16);
    }

    // ---- staging
    // K: LDS-DMA of 64 x 17 = 1088 slots (4 per thread + one more on wave 0); slot u = (row u / 17, unit u % 17), the pad unit
    //    re-reads unit 0.  V: 1024 units through registers, 4 per thread; unit u -> (key = u & 63 = tid & 63,
    //    gp = u >> 6 = 2 d-group + plane): a wave walks the keys of one (d-group, plane).
    const char* kbase = a.k + (size_t)b * a.Nk * a.pk + h * 256;
    const char* vbase = a.v + (size_t)b * a.Nk * a.pv + h * 256;
    auto issue_k = [&](int k0, int buf) {
        char* base = Ks + buf * A4_KS_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < A4KI; i++) {
            if (i == A4KI - 1 && wave != 0) break;                // slots 1024..1087: one wave
            const int u = min(tid + A4T * i, A4_KSLOTS - 1);
            const int row = (u * 3856) >> 16;                     // u / 17 for u < 1088
            const int cu = u - row * 17;
            const int kk = min(k0 + row, a.Nk - 1);               // keys past Nk: finite copies, masked to -inf below
            __builtin_amdgcn_global_load_lds((a4_gptr)(kbase + (size_t)kk * a.pk + (cu == 16 ? 0 : cu) * 16),
                                             (a4_lptr)(base + A4T * 16 * i), 16, 0, 0);
        }
    };
    // V^T position of this thread's key: pos = 32 kt + 16 s2 + 8 hh + j with t = key & 15: hh = (t >> 2) & 1,
    // j = (t & 3) + 4 (t >> 3)   (the order the S^T accumulator holds its keys)
    const int vkey = tid & 63, vt = vkey & 15;
    const int vpos2 = ((vkey & 48) + ((vt >> 2) & 1) * 8 + (vt & 3) + 4 * (vt >> 3)) * 2;
    a4_u32x4 rv[A4VI];
    auto load_v = [&](int k0) {
        const int vk = min(k0 + vkey, a.Nk - 1);
        const char* src = vbase + (size_t)vk * a.pv + wave * 16;
#pragma unroll
        for (int i = 0; i < A4VI; i++) rv[i] = *reinterpret_cast<const a4_u32x4*>(src + 4 * 16 * i);      // gp = wave + 4 i
    };
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < A4VI; i++) {
            const int gp = wave + 4 * i, g = gp >> 1, p = gp & 1;                                        // wave-uniform
            char* dst = Vt + (p * 64 + 8 * g) * A4_VROW + vpos2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = rv[i][j >> 1];
                *reinterpret_cast<uint16_t*>(dst + j * A4_VROW) = (j & 1) ? (uint16_t)(w >> 16) : (uint16_t)(w & 0xffffu);
            }
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float SCALE_LOG2E = a.scale_log2e;                    // hd^-0.5 (and the operands' power-of-two scales) folded into the exp2 argument

    const int kfrag = qi * A4_KROW + half * 32;       // this lane's K operand: row qi (+32 kb), units (2 st + half) 2 + p
    const int vfrag = qi * A4_VROW + half * 16;       // this lane's V^T operand: row d = qi (+32 db), unit 4 kb + 2 s2 + half
    const int ntiles = (a.Nk + A4K - 1) / A4K;
    issue_k(0, 0);
    load_v(0);
    store_v();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int k0 = t * A4K;
        if (t + 1 < ntiles) {
            issue_k(k0 + A4K, (t + 1) & 1);        // that buffer was last read in iteration t-1: every wave has passed a barrier since
            load_v(k0 + A4K);
        }
        const char* Kt = Ks + (t & 1) * A4_KS_BYTES;
        // two 32-key blocks, each a full online-softmax step (keeps one score tile live at a time)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            // ---- S^T = K Q^T
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                f16x8 kf[2];
#pragma unroll
                for (int p = 0; p < 2; p++)
                    kf[p] = *reinterpret_cast<const f16x8*>(Kt + kfrag + kb * 32 * A4_KROW + st * 64 + p * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[st][0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[st][1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[st][0], s, 0, 0, 0);
            }
            // ---- online softmax (keys of this lane: k0 + 32 kb + (e&3) + 8*(e>>2) + 4*half)
            if (k0 + kb * 32 + 32 > a.Nk) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (key >= a.Nk) s[e] = -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; e++) mx = fmaxf(mx, s[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);        // finite: block kb = 0 of every tile has a valid key, and m_run carries it on
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SCALE_LOG2E);
            m_run = m_new;
            // p 2^10 = exp2(s c - m c + 10) by ONE fma per element: the power of two that the fp16 planes of P need (A4_PSCALE) rides
            // in the exponent, and l accumulates the scaled values (an exact scaling: O / l is unchanged)
            const float off = __fmaf_rn(-m_new, SCALE_LOG2E, A4_PSHIFT);
            float lsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(__fmaf_rn(s[e], SCALE_LOG2E, off));
                s[e] = p;
                lsum += p;
            }
            l_run = l_run * alpha + lsum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {          // the running max moved for some query of this wave (rare after the first tiles)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int e = 0; e < 16; e++) oacc[i][e] *= alpha;
            }
            // ---- O^T += V^T P^T: split 1024 P into two fp16 planes (k-step s2 = accumulator elements 8 s2 .. 8 s2 + 7)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                a4_u32x4 pf[2];
#pragma unroll
                for (int mm = 0; mm < 4; mm++) {
                    uint32_t p0, p1;
                    fh2_split2(s[8 * s2 + 2 * mm], s[8 * s2 + 2 * mm + 1], p0, p1);
                    pf[0][mm] = p0; pf[1][mm] = p1;
                }
                const f16x8 b0 = __builtin_bit_cast(f16x8, pf[0]), b1 = __builtin_bit_cast(f16x8, pf[1]);
#pragma unroll
                for (int db = 0; db < 2; db++) {
                    f16x8 vf[2];
#pragma unroll
                    for (int p = 0; p < 2; p++)
                        vf[p] = *reinterpret_cast<const f16x8*>(Vt + vfrag + (p * 64 + db * 32) * A4_VROW + (kb * 4 + s2 * 2) * 16);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1], b0, oacc[db], 0, 0, 0);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], b1, oacc[db], 0, 0, 0);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], b0, oacc[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();                       // every wave is done with this tile's V^T image
        if (t + 1 < ntiles) store_v();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's K DMAs for tile t+1 have landed
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = a.out_mul / l_tot;                    // P and l both carry the factor 2^10: it cancels
    float amax = 0.f;
    if (q_row < a.Nq) {
        // lane (query, half) holds d = 32 db + 8 g + 4 half + (0..3): half of each plane's unit
        char* op = a.o + ((size_t)b * a.Nq + q_row) * a.po;
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {oacc[db][4 * g] * inv_l, oacc[db][4 * g + 1] * inv_l, oacc[db][4 * g + 2] * inv_l,
                                 oacc[db][4 * g + 3] * inv_l};
                amax = fh2_amax4(amax, v);
                fh2_store4(op, h * 64 + db * 32 + 8 * g + 4 * half, v);
            }
    }
    __shared__ unsigned s_amax[A4T / 64];
    fh2_publish_block(a.out_absmax, amax, s_amax);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Second form (round 3, default): V arrives like K -- by LDS-DMA into a row-major [key][256 B] image -- and its transposed MFMA
// operand is read with gfx950's ds_read_b64_tr_b16 (4 keys x 16 channels per 16-lane group, delivered column-major): no register
// staging of V (16 VGPRs live across the whole tile in the first form), no 2-byte transposition stores (32 per thread and tile).
// The registers that frees pay for a one-deep prefetch of the K and V fragments: the first form's read -> wait -> multiply order
// exposed the LDS latency eight times per 32-key block of each product, and prefetching there cost the third workgroup per CU.
// LDS image of a 64-key tile (K and V alike): row = key, 256 bytes = 16 chunks of 16 bytes in the LOGICAL order [plane 0: channel
// groups 0..7 | plane 1: channel groups 0..7] (the DMA's source address de-interleaves the fh2 row), stored at physical chunk
// c ^ swz(row), swz(row) = ((row & 3) << 2) | ((row >> 2) & 3): conflict-free for the row reads of the 32x32x16 A operand (S^T = K Q^T)
// and for the transposed reads (cdna_hip_programming.md T10, image (b)).  K double-buffered, V single-buffered (its DMA for tile t + 1
// is issued when every wave has finished tile t and has the first block's QK^T and softmax to land): 48 KB, three workgroups per CU.
#ifndef A3R_ATTN_SB
#define A3R_ATTN_SB 1
#endif
#if A3R_ATTN_SB == 1
#define B4_PIN() __builtin_amdgcn_sched_barrier(0)
#elif A3R_ATTN_SB == 2
#define B4_PIN() __builtin_amdgcn_sched_barrier(0x6)        // VALU and SALU may cross, MFMA and DS may not
#else
#define B4_PIN()
#endif
constexpr int B4_ROW = 256;
constexpr int B4_TILE = A4K * B4_ROW;          // 16 KB
constexpr int B4_LDS_BYTES = 3 * B4_TILE;      // K[2] + V
typedef short b4_s16x4 __attribute__((ext_vector_type(4)));
typedef short b4_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int b4_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ b4_s16x4 b4_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) b4_s16x4*)(p));
}

// PASSES 1 (a3r_fh2_set_passes(1)): first planes only -- q0 k0 and v0 p0 with p rounded to one fp16 -- the 16-bit operand mode.
template <int PASSES>
__global__ __launch_bounds__(A4T, 3) void attn_fh2_v2_kernel(Attn4Args a) {
    constexpr int NPL = PASSES == 1 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                // [2][64 keys][256 B]
    char* Vs = smem + 2 * B4_TILE;                  // [64 keys][256 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, half = lane >> 5;
    const int nqb = (a.Nq + A4Q - 1) / A4Q;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int group = (seq / nqb) * 8 + xcd, qb = seq - (seq / nqb) * nqb;
    if (group >= a.B * a.H) return;                 // uniform per workgroup
    const int h = group % a.H, b = group / a.H;
    const int q_row = qb * A4Q + wave * 32 + qi;
    const int q_ld = q_row < a.Nq ? q_row : a.Nq - 1;

    f16x8 qf[4][NPL];
    {
        const char* qp = a.q + ((size_t)b * a.Nq + q_ld) * a.pq + h * 256;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int p = 0; p < NPL; p++) qf[s][p] = *reinterpret_cast<const f16x8*>(qp + ((2 * s + half) * 2 + p) * 16);
    }
    // ---- DMA: slot u = tid + 256 i -> (row = (tid >> 4) + 16 i, physical chunk tid & 15); swz(row) does not depend on i
    const int drow = tid >> 4;
    const int dch = (tid & 15) ^ b4_swz(drow);                         // logical chunk this thread fetches
    const int dsrc = ((dch & 7) * 2 + (dch >> 3)) * 16;                // its byte offset in the fh2 row slice of the head
    const char* kbase = a.k + (size_t)b * a.Nk * a.pk + h * 256 + dsrc;
    const char* vbase = a.v + (size_t)b * a.Nk * a.pv + h * 256 + dsrc;
    auto issue = [&](const char* base, size_t pitch, int k0, char* buf) {
        char* dst = buf + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int kk = min(k0 + drow + 16 * i, a.Nk - 1);          // keys past Nk: finite copies (masked to -inf / multiplied by p = 0)
            __builtin_amdgcn_global_load_lds((a4_gptr)(base + (size_t)kk * pitch), (a4_lptr)(dst + 4096 * i), 16, 0, 0);
        }
    };
    // ---- fragment addresses (bytes inside a tile)
    const int ksw = b4_swz(qi);
    int koff[4][2];                                                    // K row read: row qi (+ 32 kb), logical chunk p 8 + 2 st + half
#pragma unroll
    for (int st = 0; st < 4; st++)
#pragma unroll
        for (int p = 0; p < 2; p++) koff[st][p] = qi * B4_ROW + ((((p << 3) | (st << 1) | half) ^ ksw) << 4);
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, dsub = (lane >> 4) & 1;
    int voff[2][2][2];                                                 // V transposed read [plane][d block][key group of 4]
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int g2 = 0; g2 < 2; g2++) {
                const int row = 4 * half + 8 * g2 + tq;                // (+ 16 s2 + 32 kb: immediates)
                const int ch = (p << 3) + 4 * db + 2 * dsub + (tp >> 1);
                const int sw = (tq << 2) | ((half + 2 * g2) & 3);      // = swz(row + 16 s2 + 32 kb)
                voff[p][db][g2] = row * B4_ROW + ((ch ^ sw) << 4) + 8 * (tp & 1);
            }

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float SCALE_LOG2E = a.scale_log2e;
    const int ntiles = (a.Nk + A4K - 1) / A4K;
    issue(kbase, a.pk, 0, Ks);
    issue(vbase, a.pv, 0, Vs);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                   // Q and K tile 0 have landed (V tile 0 may be in flight)
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int k0 = t * A4K;
        const bool more = t + 1 < ntiles;                              // uniform
        if (more) issue(kbase, a.pk, k0 + A4K, Ks + ((t + 1) & 1) * B4_TILE);      // that buffer was last read in tile t - 1
        const char* Kt = Ks + (t & 1) * B4_TILE;
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            // ---- S^T = K Q^T, fragments of d-step st + 1 requested before the MFMAs of d-step st
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; e++) s[e] = 0.f;
            f16x8 kf[2][NPL];
#pragma unroll
            for (int p = 0; p < NPL; p++) kf[0][p] = *reinterpret_cast<const f16x8*>(Kt + kb * 32 * B4_ROW + koff[0][p]);
#pragma unroll
            for (int st = 0; st < 4; st++) {
                if (st < 3) {
#pragma unroll
                    for (int p = 0; p < NPL; p++) kf[(st + 1) & 1][p] = *reinterpret_cast<const f16x8*>(Kt + kb * 32 * B4_ROW + koff[st + 1][p]);
                }
                B4_PIN();
                if constexpr (PASSES == 3) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[st & 1][NPL - 1], qf[st][0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[st & 1][0], qf[st][NPL - 1], s, 0, 0, 0);
                }
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[st & 1][0], qf[st][0], s, 0, 0, 0);
                B4_PIN();
            }
            // ---- online softmax, exactly as the first form
            if (k0 + kb * 32 + 32 > a.Nk) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (key >= a.Nk) s[e] = -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; e++) mx = fmaxf(mx, s[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SCALE_LOG2E);
            m_run = m_new;
            const float off = __fmaf_rn(-m_new, SCALE_LOG2E, A4_PSHIFT);
            float lsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(__fmaf_rn(s[e], SCALE_LOG2E, off));
                s[e] = p;
                lsum += p;
            }
            l_run = l_run * alpha + lsum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int e = 0; e < 16; e++) oacc[i][e] *= alpha;
            }
            if (kb == 0) {
                // V tile t: this wave's DMAs (issued at the end of tile t - 1, i.e. BEFORE K tile t + 1's) have landed, then everybody's
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // ---- O^T += V^T P^T; the transposed V fragments of the next (s2, db) are requested before the current MFMAs
            a4_u32x4 pf[2][NPL];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int mm = 0; mm < 4; mm++) {
                    uint32_t p0, p1;
                    fh2_split2(s[8 * s2 + 2 * mm], s[8 * s2 + 2 * mm + 1], p0, p1);       // (PASSES 1: the residual plane is dead code)
                    pf[s2][0][mm] = p0;
                    if constexpr (PASSES == 3) pf[s2][NPL - 1][mm] = p1;
                }
            auto vread = [&](int step, b4_s16x8 (&vf)[NPL]) {           // step = 2 s2 + db
                const int s2 = step >> 1, db = step & 1;
                const char* base = Vs + (kb * 32 + s2 * 16) * B4_ROW;
#pragma unroll
                for (int p = 0; p < NPL; p++) {
                    const b4_s16x4 lo = b4_tr_read(base + voff[p][db][0]), hi = b4_tr_read(base + voff[p][db][1]);
                    vf[p] = b4_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            };
            b4_s16x8 vf[2][NPL];
            vread(0, vf[0]);
#pragma unroll
            for (int step = 0; step < 4; step++) {
                if (step < 3) vread(step + 1, vf[(step + 1) & 1]);
                B4_PIN();
                const int s2 = step >> 1, db = step & 1;
                const f16x8 b0 = __builtin_bit_cast(f16x8, pf[s2][0]), b1 = __builtin_bit_cast(f16x8, pf[s2][NPL - 1]);
                const f16x8 v0 = __builtin_bit_cast(f16x8, vf[step & 1][0]), v1 = __builtin_bit_cast(f16x8, vf[step & 1][NPL - 1]);
                if constexpr (PASSES == 3) {
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, b0, oacc[db], 0, 0, 0);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b1, oacc[db], 0, 0, 0);
                }
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b0, oacc[db], 0, 0, 0);
                B4_PIN();
            }
        }
        // every wave is done with V tile t and K tile t; K tile t + 1 (issued at the top of this iteration) has landed for everybody
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) issue(vbase, a.pv, k0 + A4K, Vs);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = a.out_mul / l_tot;
    float amax = 0.f;
    if (q_row < a.Nq) {
        char* op = a.o + ((size_t)b * a.Nq + q_row) * a.po;
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {oacc[db][4 * g] * inv_l, oacc[db][4 * g + 1] * inv_l, oacc[db][4 * g + 2] * inv_l,
                                 oacc[db][4 * g + 3] * inv_l};
                amax = fh2_amax4(amax, v);
                fh2_store4(op, h * 64 + db * 32 + 8 * g + 4 * half, v);
            }
    }
    __shared__ unsigned s_amax[A4T / 64];
    fh2_publish_block(a.out_absmax, amax, s_amax);
}

// which kernel form a3r_attention_fh2 launches: 2 (default) or 1 (A3R_ATTN=v1 / a3r_attention_fh2_set_form)
static std::atomic<int> g_attn_form{(getenv("A3R_ATTN") && std::string(getenv("A3R_ATTN")) == "v1") ? 1 : 2};

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_attention_fh2_set_form(int form) {
    A3R_CHECK_ARG(form == 1 || form == 2, "a3r_attention_fh2_set_form: form must be 1 or 2");
    return g_attn_form.exchange(form) ;
}

extern "C" int a3r_attention_fh2(const void* q2, int ldq, const void* k2, int ldk, const void* v2, int ldv, void* o2, int ldo,
                                 int B, int H, int Nq, int Nk, const a3r_fh2_attn_range* range, void* stream) {
    A3R_CHECK_ARG(q2 && k2 && v2 && o2, "a3r_attention_fh2: null pointer");
    a3r_fh2_attn_range r = range ? *range : a3r_fh2_attn_range{};
    float* rs[4] = {&r.q_scale, &r.k_scale, &r.v_scale, &r.out_scale};
    for (float* p : rs) {
        if (*p == 0.f) *p = 1.f;
        A3R_CHECK_ARG(*p > 0.f && std::isfinite(*p), "a3r_attention_fh2: scales must be positive and finite");
    }
    A3R_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "a3r_attention_fh2: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    A3R_CHECK_ARG(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ldo >= H * 64, "a3r_attention_fh2: row strides < H*64");
    A3R_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "a3r_attention_fh2: row strides must be multiples of 8");
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(q2) | reinterpret_cast<uintptr_t>(k2) | reinterpret_cast<uintptr_t>(v2) |
                    reinterpret_cast<uintptr_t>(o2)) & 15) == 0, "a3r_attention_fh2: pointers must be 16-byte aligned");
    const bool v1 = g_attn_form.load(std::memory_order_relaxed) == 1 && fh2_passes() == 3;      // A/B switch: the register-staged V form
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fh2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fh2_v2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, B4_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fh2_v2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, B4_LDS_BYTES);
        return e;
    }));
    Attn4Args a = {static_cast<const char*>(q2), static_cast<const char*>(k2), static_cast<const char*>(v2), static_cast<char*>(o2),
                   (size_t)ldq * 4, (size_t)ldk * 4, (size_t)ldv * 4, (size_t)ldo * 4, B, H, Nq, Nk,
                   0.125f * 1.4426950408889634f / (r.q_scale * r.k_scale), r.out_scale / r.v_scale, r.out_absmax};
    const int nqb = (Nq + A4Q - 1) / A4Q, groups = B * H;
    dim3 grid(8 * ((groups + 7) / 8) * nqb);
    ProfScope prof(PK_ATTENTION_FH2, 4.0 * B * H * (double)Nq * Nk * 64, as_stream(stream));
    if (v1) hipLaunchKernelGGL(attn_fh2_kernel, grid, dim3(A4T), A4_LDS_BYTES, as_stream(stream), a);
    else if (fh2_passes() == 1) hipLaunchKernelGGL(attn_fh2_v2_kernel<1>, grid, dim3(A4T), B4_LDS_BYTES, as_stream(stream), a);
    else hipLaunchKernelGGL(attn_fh2_v2_kernel<3>, grid, dim3(A4T), B4_LDS_BYTES, as_stream(stream), a);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
