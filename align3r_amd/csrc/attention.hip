// Flash-style fp32 attention for head_dim 64 on gfx950 (no [N, N] score matrix in HBM).
//
// Replaces  attn = softmax(q @ k^T * hd^-0.5); x = attn @ v   of
//   Attention.forward      croco/models/blocks.py:105-109
//   CrossAttention.forward croco/models/blocks.py:164-168
// (q and k arrive already rotated: RoPE is fused into the producing projection, gemm.hip A3R_EPI_ROPE).
//
// All products are exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  A workgroup = 4 waves = 128 queries of one
// (batch, head); each wave owns 32 queries.  Scores are computed TRANSPOSED, S^T = K Q^T, so that a query
// is a lane (column of the accumulator) and its keys sit in the accumulator registers: the softmax
// statistics are per-lane (one cross-half exchange), and the exponentiated tile is directly the B operand
// of the second product O^T = V^T P^T -- no LDS round trip or lane movement for P.
// K/V tiles of 64 keys are staged global -> registers -> LDS (double-buffered, rows padded to 68 floats).
#include "common.h"

namespace a3r {

constexpr int AQ = 128;     // queries per workgroup
constexpr int AK = 64;      // keys per tile
constexpr int ALD = 68;     // padded LDS row (floats)
constexpr int ATTN_LDS_BYTES = 2 * 2 * AK * ALD * 4;   // 69,632 B

struct AttnArgs {
    const float *q, *k, *v;
    float* o;
    int ldq, ldk, ldv, ldo;
    int B, H, Nq, Nk;
};

__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);   // [2][AK][ALD]
    float* Vs = Ks + 2 * AK * ALD;                // [2][AK][ALD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    // XCD-aware mapping: the query blocks of one (batch, head) re-read the same K/V; workgroup ids are dealt
    // round-robin over the 8 XCDs (private L2s), so give all query blocks of a head to ONE XCD.
    const int nqb = (a.Nq + AQ - 1) / AQ;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int group = (seq / nqb) * 8 + xcd, qb = seq - (seq / nqb) * nqb;
    if (group >= a.B * a.H) return;                 // uniform per workgroup
    const int h = group % a.H, b = group / a.H;
    const int q_row = qb * AQ + wave * 32 + qi;
    const int q_ld = q_row < a.Nq ? q_row : a.Nq - 1;

    // Q fragment (B operand of S^T = K Q^T): lane holds Q[q][8*kb + 4*half + t], pre-scaled by hd^-0.5
    f32x4 qf[8];
    {
        const float* qp = a.q + ((size_t)b * a.Nq + q_ld) * a.ldq + h * 64 + 4 * half;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + kb * 8);
            qf[kb] = v * 0.125f;
        }
    }
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    const float* kbase = a.k + (size_t)b * a.Nk * a.ldk + h * 64 + sc4;
    const float* vbase = a.v + (size_t)b * a.Nk * a.ldv + h * 64 + sc4;
    f32x4 rk[4], rv[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int key = k0 + srow + 16 * i;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rk[i] = z; rv[i] = z;
            if (key < a.Nk) {
                rk[i] = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * a.ldk);
                rv[i] = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * a.ldv);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = srow + 16 * i;
            *reinterpret_cast<f32x4*>(Ks + (buf * AK + row) * ALD + sc4) = rk[i];
            *reinterpret_cast<f32x4*>(Vs + (buf * AK + row) * ALD + sc4) = rv[i];
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float LOG2E = 1.4426950408889634f;

    const int ntiles = (a.Nk + AK - 1) / AK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        const int buf = t & 1, k0 = t * AK;
        if (t + 1 < ntiles) load_tile(k0 + AK);
        // ---- S^T = K Q^T
        f32x16 s[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) s[i][e] = 0.f;
        const float* Kb = Ks + (buf * AK + qi) * ALD + 4 * half;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            const f32x4 k0v = *reinterpret_cast<const f32x4*>(Kb + kb * 8);
            const f32x4 k1v = *reinterpret_cast<const f32x4*>(Kb + 32 * ALD + kb * 8);
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0v[tt], qf[kb][tt], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1v[tt], qf[kb][tt], s[1], 0, 0, 0);
            }
        }
        // ---- online softmax (keys of this lane: kt*32 + (e&3) + 8*(e>>2) + 4*half)
        if (k0 + AK > a.Nk) {
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = k0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (key >= a.Nk) s[kt][e] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) mx = fmaxf(mx, s[kt][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        m_run = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f((s[kt][e] - m_new) * LOG2E);
                s[kt][e] = p;
                lsum += p;
            }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) oacc[i][e] *= alpha;
        // ---- O^T += V^T P^T
        const float* Vb = Vs + buf * AK * ALD + qi;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const float v0 = Vb[key * ALD];
                const float v1 = Vb[key * ALD + 32];
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[kt][e], oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[kt][e], oacc[1], 0, 0, 0);
            }
        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.f / l_tot;
    if (q_row < a.Nq) {
        float* op = a.o + ((size_t)b * a.Nq + q_row) * a.ldo + h * 64 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v = {oacc[dt][4 * g] * inv_l, oacc[dt][4 * g + 1] * inv_l, oacc[dt][4 * g + 2] * inv_l,
                           oacc[dt][4 * g + 3] * inv_l};
                *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * g) = v;
            }
    }
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                             int ldo, int B, int H, int Nq, int Nk, void* stream) {
    A3R_CHECK_ARG(q && k && v && o, "a3r_attention: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "a3r_attention: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    A3R_CHECK_ARG(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ldo >= H * 64, "a3r_attention: row strides < H*64");
    A3R_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "a3r_attention: row strides must be multiples of 4");
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                    reinterpret_cast<uintptr_t>(o)) & 15) == 0, "a3r_attention: pointers must be 16-byte aligned");
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    ATTN_LDS_BYTES); }));
    AttnArgs a = {q, k, v, o, ldq, ldk, ldv, ldo, B, H, Nq, Nk};
    const int nqb = (Nq + AQ - 1) / AQ, groups = B * H;
    dim3 grid(8 * ((groups + 7) / 8) * nqb);
    ProfScope prof(PK_ATTENTION, 4.0 * B * H * (double)Nq * Nk * 64, as_stream(stream));
    hipLaunchKernelGGL(attn_kernel, grid, dim3(256), ATTN_LDS_BYTES, as_stream(stream), a);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
