// Pieces shared by the MFMA GEMM kernels of liba3r (gemm.hip: exact-fp32 MFMA; gemm_bf3.hip: split-bf16 MFMA):
// launch arguments, the fused epilogue (bias / GELU / ReLU / residuals / RoPE-2D / pixel-shuffle) and argument checks.
#pragma once
#include "common.h"
#include "bf3.h"

namespace a3r {

struct GroupPtrs {
    const float* A;
    const float* Wt;        // [N, K]
    float* C;
    const float* bias;
    const float* resid;
    const float* resid2;
};

struct GemmArgs {
    GroupPtrs grp[4];
    int groups, tiles_per_group;
    int lda, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    a3r_epilogue epi;       // pointers inside are ignored (taken from grp[]) except the rope tables
    // implicit conv (AMODE 1): A = x [B, H, W, Cin]
    int cH, cW, cCin, cHo, cWo, cStride;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// Epilogue over the 32x32 MFMA accumulators of one wave: acc[i][j] covers rows m0 + wrow0 + 32 i .. and columns
// n0 + wcol0 + 32 j ..; element e of a lane sits at row (e&3) + 8 (e>>2) + 4 (lane>>5), column lane&31.
template <int TM, int TN, bool FULL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const GroupPtrs& P, f32x16 (&acc)[TM][TN], int m0, int n0,
                                              int wrow0, int wcol0, int lane) {
    const a3r_epilogue& ep = g.epi;
    const int half = lane >> 5, lcol = lane & 31;
    const int epi = ep.epi;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int colbase = n0 + wcol0 + j * 32;
        const int col = colbase + lcol;
        const bool col_ok = FULL || col < g.N;
        const float bias = (P.bias && col_ok) ? P.bias[epi == A3R_EPI_PIXSHUF ? col % ep.ps_cout : col] : 0.f;
        const bool do_rope = epi == A3R_EPI_ROPE && colbase < ep.rope_cols;   // wave-uniform (rope_cols % 64 == 0)
        const bool rope_x = (colbase & 32) != 0;                               // second half of the head rotates with x
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                float v = acc[i][j][e] + bias;
                if (do_rope) {
                    // pairs (d, d+16) inside each 32-wide half of the head (RoPE2D pos_embed.py:130-157)
                    const float other = __shfl_xor(v, 16);
                    const int tok = row % ep.tokens_per_image;
                    const int py = tok / ep.grid_w, px = tok - py * ep.grid_w;
                    const int p = rope_x ? px : py;
                    const float c = ep.rope_cos[p * 16 + (lcol & 15)], s = ep.rope_sin[p * 16 + (lcol & 15)];
                    v = (lcol & 16) ? v * c + other * s : v * c - other * s;
                }
                if ((FULL || row < g.M) && col_ok) {
                    if (epi == A3R_EPI_GELU) v = gelu_erf(v);
                    else if (epi == A3R_EPI_RELU) v = fmaxf(v, 0.f);
                    else if (epi == A3R_EPI_RESID) v = P.resid[(size_t)row * g.ldc + col] + v;
                    else if (epi == A3R_EPI_RESID2) v = P.resid[(size_t)row * g.ldc + col] + P.resid2[(size_t)row * g.ldc + col] + v;
                    if (epi == A3R_EPI_PIXSHUF) {
                        const int s = ep.ps_s, hw = ep.ps_h * ep.ps_w;
                        const int b = row / hw, rem = row - b * hw;
                        const int y = rem / ep.ps_w, x = rem - y * ep.ps_w;
                        const int tap = col / ep.ps_cout, co = col - tap * ep.ps_cout;
                        const int dy = tap / s, dx = tap - dy * s;
                        const size_t opix = ((size_t)b * ep.ps_h * s + (y * s + dy)) * (ep.ps_w * s) + (x * s + dx);
                        P.C[opix * ep.ps_cout + co] = v;
                    } else {
                        P.C[(size_t)row * g.ldc + col] = v;
                    }
                }
            }
        }
    }
}

// The same epilogue over 16x16 accumulators (v_mfma_f32_16x16x32_bf16): acc[i][j] covers rows m0 + wrow0 + 16 i .. and
// columns n0 + wcol0 + 16 j ..; element e of a lane sits at row 4 (lane>>4) + e, column lane&15.  wcol0 % 32 == 0 and TN is
// even, so the RoPE partner column (d +- 16 inside a 32-wide half of the head) is the same lane's element of tile j ^ 1.
// bf3 outputs (out_bf3: instead of the fp32 store; aux_bf3: in addition to it, optionally through a ReLU): neighbouring
// lanes (columns c, c+1) trade half of their four rows, so that each lane owns two rows of a column PAIR and stores one
// packed dword per plane and row (the even lane rows 0-1 of its quad, the odd lane rows 2-3).
// EPI >= 0 / O3 >= 0: the epilogue kind / the presence of a bf3 output fixed at compile time (straight-line code for the hot
// combinations); -1: decided at run time (generic body).
template <int TM, int TN, bool FULL, int EPI, int O3>
__device__ __forceinline__ void gemm_epilogue16_body(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][TN], int m0, int n0,
                                                     int wrow0, int wcol0, int lane) {
    static_assert(TN % 2 == 0, "pairs of 16-wide tiles");
    const a3r_epilogue& ep = g.epi;
    const int quad = lane >> 4, lcol = lane & 15;
    const int epi = EPI >= 0 ? EPI : ep.epi;
    const bool odd = lane & 1;
    const bool only3 = O3 == 0 ? false : ep.out_bf3 != 0;                 // bf3 INSTEAD of the fp32 store
    char* out3 = O3 == 0 ? nullptr : (only3 ? reinterpret_cast<char*>(P.C) : static_cast<char*>(ep.aux_bf3));
    const bool relu3 = !only3 && ep.aux_relu;
    const size_t pitch3 = (size_t)g.N * 6;
    const int pair3 = only3 && ep.out_pair;                               // out_bf3 in the row-pair layout (bf3.h)
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int colbase = n0 + wcol0 + j * 16;
        const int col = colbase + lcol;
        const bool col_ok = FULL || col < g.N;
        const float bias = (P.bias && col_ok) ? P.bias[epi == A3R_EPI_PIXSHUF ? col % ep.ps_cout : col] : 0.f;
        const float bias_o = (epi == A3R_EPI_ROPE && P.bias) ? P.bias[min(col ^ 16, g.N - 1)] : 0.f;   // partner column's bias
        const bool do_rope = epi == A3R_EPI_ROPE && colbase < ep.rope_cols;   // wave-uniform (rope_cols % 64 == 0)
        const bool rope_x = (colbase & 32) != 0;                               // second half of the head rotates with x
        const bool second = (colbase & 16) != 0;                               // d in [16, 32) of the half: partner is d - 16
#pragma unroll
        for (int i = 0; i < TM; i++) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = m0 + wrow0 + i * 16 + quad * 4 + e;
                v[e] = acc[i][j][e] + bias;
                if (do_rope) {
                    // pairs (d, d+16) inside each 32-wide half of the head (RoPE2D pos_embed.py:130-157)
                    const float other = acc[i][j ^ 1][e] + bias_o;
                    const int tok = row % ep.tokens_per_image;
                    const int py = tok / ep.grid_w, px = tok - py * ep.grid_w;
                    const int p = rope_x ? px : py;
                    const float c = ep.rope_cos[p * 16 + lcol], s = ep.rope_sin[p * 16 + lcol];
                    v[e] = second ? v[e] * c + other * s : v[e] * c - other * s;
                }
                const bool ok = (FULL || row < g.M) && col_ok;
                if (epi == A3R_EPI_GELU) v[e] = gelu_erf(v[e]);
                else if (epi == A3R_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                else if (epi == A3R_EPI_RESID) { if (ok) v[e] = P.resid[(size_t)row * g.ldc + col] + v[e]; }
                else if (epi == A3R_EPI_RESID2) { if (ok) v[e] = P.resid[(size_t)row * g.ldc + col] + P.resid2[(size_t)row * g.ldc + col] + v[e]; }
                if (ok && !only3) {
                    if (epi == A3R_EPI_PIXSHUF) {
                        const int s = ep.ps_s, hw = ep.ps_h * ep.ps_w;
                        const int b = row / hw, rem = row - b * hw;
                        const int y = rem / ep.ps_w, x = rem - y * ep.ps_w;
                        const int tap = col / ep.ps_cout, co = col - tap * ep.ps_cout;
                        const int dy = tap / s, dx = tap - dy * s;
                        const size_t opix = ((size_t)b * ep.ps_h * s + (y * s + dy)) * (ep.ps_w * s) + (x * s + dx);
                        P.C[opix * ep.ps_cout + co] = v[e];
                    } else {
                        P.C[(size_t)row * g.ldc + col] = v[e];
                    }
                }
            }
            if (out3) {                                       // wave-uniform
                if (relu3) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
                }
                const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];
                const float r0 = dpp_xor1(s0), r1 = dpp_xor1(s1);
                const float a0 = odd ? r0 : v[0], b0 = odd ? v[2] : r0;      // (left, right column) of this lane's first row
                const float a1 = odd ? r1 : v[1], b1 = odd ? v[3] : r1;
                const int row0 = m0 + wrow0 + i * 16 + quad * 4 + (odd ? 2 : 0);
                const int c0 = col & ~1;                                      // N % 8 == 0: a column pair is in or out together
                char* d = out3 + bf3_row_offset(row0, g.N, pair3) + bf3_k_offset(c0, pair3) + (c0 & 7) * 2;      // row0 is even
                const size_t next_row = pair3 ? 192 : pitch3;
                uint32_t p0, p1, p2;
                if (col_ok && (FULL || row0 < g.M)) {
                    bf3_split2(a0, b0, p0, p1, p2);
                    *reinterpret_cast<uint32_t*>(d) = p0;
                    *reinterpret_cast<uint32_t*>(d + 16) = p1;
                    *reinterpret_cast<uint32_t*>(d + 32) = p2;
                }
                if (col_ok && (FULL || row0 + 1 < g.M)) {
                    bf3_split2(a1, b1, p0, p1, p2);
                    *reinterpret_cast<uint32_t*>(d + next_row) = p0;
                    *reinterpret_cast<uint32_t*>(d + next_row + 16) = p1;
                    *reinterpret_cast<uint32_t*>(d + next_row + 32) = p2;
                }
            }
        }
    }
}

template <int TM, int TN, bool FULL>
__device__ __forceinline__ void gemm_epilogue16(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][TN], int m0, int n0,
                                                int wrow0, int wcol0, int lane) {
    const a3r_epilogue& ep = g.epi;
    const bool has3 = ep.out_bf3 || ep.aux_bf3;                            // wave-uniform dispatch, once per call
    if (!has3) {
        if (ep.epi == A3R_EPI_RESID) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_RESID, 0>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else if (ep.epi == A3R_EPI_NONE) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_NONE, 0>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else gemm_epilogue16_body<TM, TN, FULL, -1, 0>(g, P, acc, m0, n0, wrow0, wcol0, lane);
    } else {
        if (ep.epi == A3R_EPI_ROPE) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_ROPE, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else if (ep.epi == A3R_EPI_GELU) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_GELU, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else if (ep.epi == A3R_EPI_RESID) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_RESID, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else if (ep.epi == A3R_EPI_RESID2) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_RESID2, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else if (ep.epi == A3R_EPI_RELU) gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_RELU, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);
        else gemm_epilogue16_body<TM, TN, FULL, A3R_EPI_NONE, 1>(g, P, acc, m0, n0, wrow0, wcol0, lane);      // (PIXSHUF has no bf3 form)
    }
}

static inline int check_epilogue(const a3r_epilogue* e, int M, int N, const char* who, bool bf3_kernel = false) {
    if (!e) return A3R_OK;
    if (e->out_bf3) {
        A3R_CHECK_ARG(bf3_kernel, "%s: out_bf3 is only available on the bf3 kernels", who);
        A3R_CHECK_ARG(N % 8 == 0 && (e->epi == A3R_EPI_NONE || e->epi == A3R_EPI_GELU || e->epi == A3R_EPI_RELU || e->epi == A3R_EPI_ROPE),
                      "%s: out_bf3 needs N %% 8 == 0 and a NONE / GELU / RELU / ROPE epilogue", who);
        A3R_CHECK_ARG(!e->aux_bf3, "%s: out_bf3 and aux_bf3 are exclusive", who);
        A3R_CHECK_ARG(!e->out_pair || N % 32 == 0, "%s: out_pair needs N %% 32 == 0", who);
    } else {
        A3R_CHECK_ARG(!e->out_pair, "%s: out_pair without out_bf3", who);
    }
    if (e->aux_bf3) {
        A3R_CHECK_ARG(bf3_kernel, "%s: aux_bf3 is only available on the bf3 kernels", who);
        A3R_CHECK_ARG(N % 8 == 0 && e->epi != A3R_EPI_PIXSHUF && (reinterpret_cast<uintptr_t>(e->aux_bf3) & 15) == 0,
                      "%s: aux_bf3 needs N %% 8 == 0, a 16-byte aligned buffer and no PIXSHUF", who);
    }
    A3R_CHECK_ARG(e->epi >= A3R_EPI_NONE && e->epi <= A3R_EPI_PIXSHUF, "%s: unknown epilogue %d", who, e->epi);
    if (e->epi == A3R_EPI_ROPE)
        A3R_CHECK_ARG(e->rope_cols % 64 == 0 && e->rope_cols <= N && e->tokens_per_image > 0 && e->grid_w > 0 &&
                          e->tokens_per_image % e->grid_w == 0 && e->rope_cos && e->rope_sin,
                      "%s: bad ROPE epilogue (rope_cols=%d tokens=%d grid_w=%d)", who, e->rope_cols,
                      e->tokens_per_image, e->grid_w);
    if (e->epi == A3R_EPI_PIXSHUF)
        A3R_CHECK_ARG(e->ps_s > 0 && e->ps_cout > 0 && N == e->ps_s * e->ps_s * e->ps_cout && e->ps_h > 0 && e->ps_w > 0 &&
                          M % (e->ps_h * e->ps_w) == 0,
                      "%s: bad PIXSHUF epilogue", who);
    return A3R_OK;
}

static inline int check_group(const GroupPtrs& p, int epi, const char* who) {
    A3R_CHECK_ARG(p.A && p.Wt && p.C, "%s: null pointer", who);
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.Wt)) & 15) == 0,
                  "%s: x and w must be 16-byte aligned", who);
    if (epi == A3R_EPI_RESID || epi == A3R_EPI_RESID2) A3R_CHECK_ARG(p.resid, "%s: RESID epilogue without resid", who);
    if (epi == A3R_EPI_RESID2) A3R_CHECK_ARG(p.resid2, "%s: RESID2 epilogue without resid2", who);
    return A3R_OK;
}

}  // namespace a3r
