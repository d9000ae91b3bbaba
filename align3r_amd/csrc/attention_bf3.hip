// Flash-style attention for head_dim 64 on the bf16 matrix cores, fp32-accurate ("bf3" operands, bf3.h).
//
// Replaces  attn = softmax(q @ k^T * hd^-0.5); x = attn @ v   of
//   Attention.forward      croco/models/blocks.py:105-109
//   CrossAttention.forward croco/models/blocks.py:164-168
// like attention.hip, but both products run as six exact bf16 x bf16 MFMA passes over three-plane splits:
//   * q, k, v arrive in bf3 form straight from the projection GEMM (RoPE + out_bf3 epilogue, gemm_bf3.hip);
//   * S^T = K Q^T  (v_mfma_f32_32x32x16_bf16; a lane's K / Q operand is one 16-byte bf3 unit per plane);
//   * online softmax in fp32 on the accumulators (a query is a lane, its keys sit in the registers);
//   * P is split into three bf16 planes in registers (exactly: p = p0 + p1 + p2) and is directly the B operand of
//     O^T = V^T P^T; V^T needs 8 consecutive KEYS per lane, so the V tile is staged through LDS transposed, with the
//     keys of each 16-key step permuted to the order the S^T accumulator holds them (the contraction order is free);
//   * O is written in bf3 form, the input format of the output projection GEMM.
// A workgroup = 4 waves = 128 queries of one (batch, head), each wave 32 queries; two workgroups share a CU.  K tiles of 64 keys are prefetched by
// LDS-DMA into a double buffer, V tiles global -> registers during the tile's MFMAs and scattered to LDS between two
// barriers; inside a tile each 32-key block is a complete online-softmax step (one score accumulator live).
// LDS images are conflict-free for ds_read_b128 by padding: K rows are 24 + 1 units, V^T rows 8 + 1 units (odd strides).
#include "common.h"
#include "bf3.h"
#include "fh2.h"

namespace a3r {

#ifndef A3_WAVES
#define A3_WAVES 4                          // waves per workgroup: 4 (two workgroups per CU, independent phases) or 8
#endif
constexpr int A3T = 64 * A3_WAVES;          // threads per workgroup
constexpr int A3Q = 32 * A3_WAVES;          // queries per workgroup (each wave 32)
constexpr int A3KI = (1600 + A3T - 1) / A3T;   // K DMA slots per thread (the last one covers only the first wave(s))
constexpr int A3VI = 1536 / A3T;            // V units per thread
constexpr int A3K = 64;                     // keys per tile
// LDS images are padded by one 16-byte unit per row (K: 24 + 1 units, V^T: 8 + 1): an odd row stride in units makes every
// ds_read_b128 of an MFMA operand conflict-free with plain immediate offsets (no per-lane swizzle arithmetic).
constexpr int A3_KROW = 25 * 16;            // K row: 24 units + 1 pad unit (the pad is DMA'd too: LDS-DMA writes lane-linear)
constexpr int A3_KS_BYTES = A3K * A3_KROW;  // one K tile (two of them: LDS-DMA double buffer)
constexpr int A3_VROW = 9 * 16;             // V^T row: 64 keys x 2 B + 1 pad unit
constexpr int A3_VT_BYTES = 3 * 64 * A3_VROW;   // V^T tile: 3 planes x 64 d
constexpr int A3_LDS_BYTES = 2 * A3_KS_BYTES + A3_VT_BYTES;   // 78,848 B

struct Attn3Args {
    const char *q, *k, *v;
    char* o;
    size_t pq, pk, pv, po;                  // row pitches in bytes (6 x leading dimension)
    int B, H, Nq, Nk;
    int ldo, out_pair;                      // o: leading dimension in elements; row-pair layout (bf3.h) for a GEMM-only consumer
    int out_fh2;                            // o in fh2 form (fh2.h, scale 1, plain rows) for an a3r_linear_fh2 output projection
};

typedef const __attribute__((address_space(1))) void* a3_gptr;
typedef __attribute__((address_space(3))) void* a3_lptr;

template <int NP>
__global__ __launch_bounds__(A3T, A3_WAVES == 4 ? 2 : 1) void attn_bf3_kernel(Attn3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                // [2][64 rows][24 units]
    char* Vt = smem + 2 * A3_KS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, half = lane >> 5;
    // XCD-aware mapping (as attention.hip): all query blocks of one (batch, head) land on ONE XCD
    const int nqb = (a.Nq + A3Q - 1) / A3Q;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int group = (seq / nqb) * 8 + xcd, qb = seq - (seq / nqb) * nqb;
    if (group >= a.B * a.H) return;                 // uniform per workgroup
    const int h = group % a.H, b = group / a.H;
    const int q_row = qb * A3Q + wave * 32 + qi;
    const int q_ld = q_row < a.Nq ? q_row : a.Nq - 1;

    // Q operand (B of S^T = K Q^T): lane (query, half) holds, per 16-deep d-step s and plane p, unit (2 s + half) 3 + p
    bf16x8 qf[4][3];
    {
        const char* qp = a.q + ((size_t)b * a.Nq + q_ld) * a.pq + h * 384;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int p = 0; p < 3; p++) qf[s][p] = *reinterpret_cast<const bf16x8*>(qp + ((2 * s + half) * 3 + p) * 16);
    }

    // ---- staging
    // K: LDS-DMA of 64 x 25 = 1600 slots (3 per thread + one more on wave 0); slot u = (row u / 25, unit u % 25), the pad
    //    unit re-reads unit 0.  V: 1536 units through registers, 3 per thread; unit u -> (key = u & 63 = tid & 63,
    //    gp = u >> 6 = 3 kgroup + plane): a wave walks the keys of one (kgroup, plane).
    const char* kbase = a.k + (size_t)b * a.Nk * a.pk + h * 384;
    const char* vbase = a.v + (size_t)b * a.Nk * a.pv + h * 384;
    int krow_[A3KI], kcu_[A3KI];
#pragma unroll
    for (int i = 0; i < A3KI; i++) {
        const int u = min(tid + A3T * i, 1599);
        krow_[i] = u / 25;
        const int cu = u - krow_[i] * 25;
        kcu_[i] = (cu == 24 ? 0 : cu) * 16;
    }
    auto issue_k = [&](int k0, int buf) {
        char* base = Ks + buf * A3_KS_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < A3KI; i++) {
            if (i == A3KI - 1 && wave != 0) break;                // slots 1536..1599: one wave
            const int kk = min(k0 + krow_[i], a.Nk - 1);          // keys past Nk: finite copies, masked to -inf below
            __builtin_amdgcn_global_load_lds((a3_gptr)(kbase + (size_t)kk * a.pk + kcu_[i]), (a3_lptr)(base + A3T * 16 * i), 16, 0, 0);
        }
    };
    // V^T position of this thread's key: pos = 32 kt + 16 s2 + 8 hh + j with t = key & 15: hh = (t >> 2) & 1,
    // j = (t & 3) + 4 (t >> 3)   (the order the S^T accumulator holds its keys)
    const int vkey = tid & 63, vt = vkey & 15;
    const int vpos2 = ((vkey & 48) + ((vt >> 2) & 1) * 8 + (vt & 3) + 4 * (vt >> 3)) * 2;
    u32x4 rv[A3VI];
    auto load_v = [&](int k0) {
        const int vk = min(k0 + vkey, a.Nk - 1);
        const char* src = vbase + (size_t)vk * a.pv + wave * 16;
#pragma unroll
        for (int i = 0; i < A3VI; i++) rv[i] = *reinterpret_cast<const u32x4*>(src + A3_WAVES * 16 * i);      // gp = wave + A3_WAVES i
    };
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < A3VI; i++) {
            const int gp = wave + A3_WAVES * i, g = gp / 3, p = gp - 3 * g;                        // wave-uniform
            char* dst = Vt + (p * 64 + 8 * g) * A3_VROW + vpos2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = rv[i][j >> 1];
                *reinterpret_cast<uint16_t*>(dst + j * A3_VROW) = (j & 1) ? (uint16_t)(w >> 16) : (uint16_t)(w & 0xffffu);
            }
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float SCALE_LOG2E = 0.125f * 1.4426950408889634f;     // hd^-0.5 folded into the exp2 argument

    const int kfrag = qi * A3_KROW + half * 48;       // this lane's K operand: row qi (+32 kb), units (2 st + half) 3 + p
    const int vfrag = qi * A3_VROW + half * 16;       // this lane's V^T operand: row d = qi (+32 db), unit 4 kb + 2 s2 + half
    const int ntiles = (a.Nk + A3K - 1) / A3K;
    issue_k(0, 0);
    load_v(0);
    store_v();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int k0 = t * A3K;
        if (t + 1 < ntiles) {
            issue_k(k0 + A3K, (t + 1) & 1);        // that buffer was last read in iteration t-1: every wave has passed a barrier since
            load_v(k0 + A3K);
        }
        const char* Kt = Ks + (t & 1) * A3_KS_BYTES;
        // two 32-key blocks, each a full online-softmax step (keeps one score tile live at a time)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            // ---- S^T = K Q^T
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                bf16x8 kf[3];
#pragma unroll
                for (int p = 0; p < 3; p++)
                    kf[p] = *reinterpret_cast<const bf16x8*>(Kt + kfrag + kb * 32 * A3_KROW + st * 96 + p * 16);
                if (NP == 6) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[st][0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[st][1], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[st][2], s, 0, 0, 0);
                }
                if (NP >= 3) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[st][0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[st][1], s, 0, 0, 0);
                }
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[st][0], s, 0, 0, 0);
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
            float lsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f((s[e] - m_new) * SCALE_LOG2E);
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
            // ---- O^T += V^T P^T: split P into planes (k-step s2 = accumulator elements 8 s2 .. 8 s2 + 7)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                u32x4 pf[3];
#pragma unroll
                for (int mm = 0; mm < 4; mm++) {
                    uint32_t p0, p1, p2;
                    bf3_split2(s[8 * s2 + 2 * mm], s[8 * s2 + 2 * mm + 1], p0, p1, p2);
                    pf[0][mm] = p0; pf[1][mm] = p1; pf[2][mm] = p2;
                }
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, pf[0]), b1 = __builtin_bit_cast(bf16x8, pf[1]),
                             b2 = __builtin_bit_cast(bf16x8, pf[2]);
#pragma unroll
                for (int db = 0; db < 2; db++) {
                    bf16x8 vf[3];
#pragma unroll
                    for (int p = 0; p < 3; p++)
                        vf[p] = *reinterpret_cast<const bf16x8*>(Vt + vfrag + (p * 64 + db * 32) * A3_VROW + (kb * 4 + s2 * 2) * 16);
                    if (NP == 6) {
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], b0, oacc[db], 0, 0, 0);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], b1, oacc[db], 0, 0, 0);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], b2, oacc[db], 0, 0, 0);
                    }
                    if (NP >= 3) {
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], b0, oacc[db], 0, 0, 0);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], b1, oacc[db], 0, 0, 0);
                    }
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], b0, oacc[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();                       // every wave is done with this tile's V^T image
        if (t + 1 < ntiles) store_v();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's K DMAs for tile t+1 have landed
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.f / l_tot;
    if (q_row < a.Nq) {
        // lane (query, half) holds d = 32 db + 8 g + 4 half + (0..3): half a bf3 unit per plane
        char* op = a.out_fh2 ? a.o + ((size_t)b * a.Nq + q_row) * fh2_row_bytes(a.ldo)
                             : a.o + bf3_row_offset((long)b * a.Nq + q_row, a.ldo, a.out_pair);
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {oacc[db][4 * g] * inv_l, oacc[db][4 * g + 1] * inv_l, oacc[db][4 * g + 2] * inv_l,
                                 oacc[db][4 * g + 3] * inv_l};
                if (a.out_fh2) fh2_store4(op, h * 64 + db * 32 + 8 * g + 4 * half, v);
                else bf3_store4(op, h * 64 + db * 32 + 8 * g + 4 * half, v, a.out_pair);
            }
    }
}

}  // namespace a3r
using namespace a3r;

static int attention_bf3_impl(const void* q3, int ldq, const void* k3, int ldk, const void* v3, int ldv, void* o3, int ldo,
                              int B, int H, int Nq, int Nk, int out_pair, int out_fh2, void* stream) {
    A3R_CHECK_ARG(q3 && k3 && v3 && o3, "a3r_attention_bf3: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "a3r_attention_bf3: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    A3R_CHECK_ARG(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ldo >= H * 64, "a3r_attention_bf3: row strides < H*64");
    A3R_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "a3r_attention_bf3: row strides must be multiples of 8");
    A3R_CHECK_ARG(!out_pair || ldo % 32 == 0, "a3r_attention_bf3: the row-pair output layout needs ldo %% 32 == 0");
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(q3) | reinterpret_cast<uintptr_t>(k3) | reinterpret_cast<uintptr_t>(v3) |
                    reinterpret_cast<uintptr_t>(o3)) & 15) == 0, "a3r_attention_bf3: pointers must be 16-byte aligned");
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] {
        hipError_t e = hipSuccess;
        const void* kerns[3] = {reinterpret_cast<const void*>(&attn_bf3_kernel<6>), reinterpret_cast<const void*>(&attn_bf3_kernel<3>),
                                reinterpret_cast<const void*>(&attn_bf3_kernel<1>)};
        for (const void* k : kerns)
            if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, A3_LDS_BYTES);
        return e;
    }));
    Attn3Args a = {static_cast<const char*>(q3), static_cast<const char*>(k3), static_cast<const char*>(v3), static_cast<char*>(o3),
                   (size_t)ldq * 6, (size_t)ldk * 6, (size_t)ldv * 6, (size_t)ldo * 6, B, H, Nq, Nk, ldo, out_pair ? 1 : 0, out_fh2 ? 1 : 0};
    const int nqb = (Nq + A3Q - 1) / A3Q, groups = B * H;
    dim3 grid(8 * ((groups + 7) / 8) * nqb);
    ProfScope prof(PK_ATTENTION_BF3, 4.0 * B * H * (double)Nq * Nk * 64, as_stream(stream));
    const int np = bf3_products();       // process-wide arithmetic mode (a3r_bf3_set_products)
    if (np == 6) hipLaunchKernelGGL(attn_bf3_kernel<6>, grid, dim3(A3T), A3_LDS_BYTES, as_stream(stream), a);
    else if (np == 3) hipLaunchKernelGGL(attn_bf3_kernel<3>, grid, dim3(A3T), A3_LDS_BYTES, as_stream(stream), a);
    else hipLaunchKernelGGL(attn_bf3_kernel<1>, grid, dim3(A3T), A3_LDS_BYTES, as_stream(stream), a);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_attention_bf3(const void* q3, int ldq, const void* k3, int ldk, const void* v3, int ldv, void* o3, int ldo,
                                 int B, int H, int Nq, int Nk, int out_pair, void* stream) {
    return attention_bf3_impl(q3, ldq, k3, ldk, v3, ldv, o3, ldo, B, H, Nq, Nk, out_pair, 0, stream);
}

extern "C" int a3r_attention_bf3_fh2out(const void* q3, int ldq, const void* k3, int ldk, const void* v3, int ldv, void* o2, int ldo,
                                        int B, int H, int Nq, int Nk, void* stream) {
    return attention_bf3_impl(q3, ldq, k3, ldk, v3, ldv, o2, ldo, B, H, Nq, Nk, 0, 1, stream);
}
