// Pieces shared by the MFMA GEMM kernels of liba3r (gemm.hip: exact-fp32 MFMA; gemm_bf3.hip: split-bf16 MFMA):
// launch arguments, the fused epilogue (bias / GELU / ReLU / residuals / RoPE-2D / pixel-shuffle) and argument checks.
#pragma once
#include "common.h"
#include "bf3.h"
#include "fh2.h"

// Between a wave's writes of its private LDS image and its own reads of it (and back): the LDS pipeline executes one wave's DS
// instructions in program order, so only the COMPILER has to be kept from moving them across each other (wave_barrier); with
// A3R_EPI_WAIT the wave also drains lgkmcnt first (the form of rounds 1-2).
#ifdef A3R_EPI_WAIT
#define A3R_EPI_FENCE() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define A3R_EPI_FENCE() __builtin_amdgcn_wave_barrier()
#endif

namespace a3r {

struct GroupPtrs {
    const float* A;
    const float* Wt;        // [N, K]
    float* C;
    const float* bias;
    const float* resid;
    const float* resid2;
    // fh2 kernels: the power of two the out_fh2 / aux_fh2 output is stored with, and the word receiving max |out_scale * value| (or null)
    float out_scale;
    unsigned* out_absmax;
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
    int direct_epilogue;    // A/B switch (env A3R_BF3_DIRECT_EPI): keep the accumulator-layout epilogue instead of the LDS-staged one
};

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU default, blocks.py:66) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, the size of an fp32 rounding of erf): q = 0.5 (1 - erf|z|) = 0.5 t (a1 + t (a2 + ...)) exp(-z^2),
// t = 1 / (1 + p |z|); gelu = x (1 - q) for x >= 0 and x q for x < 0 -- no cancellation in the negative tail, one rcp + one exp2
// and seven fma instead of the device library's branching erff (which cost +10 % on the fc1 GEMM, tools/bench_epi_fh2.py).
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, z, 1.f));        // v_rcp_f32 (1 ulp)
    float p = __fmaf_rn(t, 1.061405429f, -1.453152027f);
    p = __fmaf_rn(t, p, 1.421413741f);
    p = __fmaf_rn(t, p, -0.284496736f);
    p = __fmaf_rn(t, p, 0.254829592f);
    const float q = 0.5f * t * p * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // v_exp_f32: flushes below 2^-126, where q is 0 to fp32 anyway
    return x >= 0.f ? __fmaf_rn(-x, q, x) : x * q;
}

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
                else if (epi == A3R_EPI_RESID) { if (ep.relu_acc) v[e] = fmaxf(v[e], 0.f); if (ok) v[e] = P.resid[(size_t)row * g.ldc + col] + v[e]; }
                else if (epi == A3R_EPI_RESID2) { if (ep.relu_acc) v[e] = fmaxf(v[e], 0.f); if (ok) v[e] = P.resid[(size_t)row * g.ldc + col] + P.resid2[(size_t)row * g.ldc + col] + v[e]; }
                if ((epi == A3R_EPI_RESID || epi == A3R_EPI_RESID2) && ep.relu_out) v[e] = fmaxf(v[e], 0.f);
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

// ---- the same epilogue staged through LDS (the stage ring is idle by then).  In the accumulator layout a lane holds ONE column
// of four rows, so the direct form above issues 4-byte loads / stores that touch four 64-byte row segments per instruction -- at
// K = 1024 the residual loads, fp32 stores and especially the 4-byte bf3 pieces cost 17-26 % of the whole GEMM (tools/bench_epi.py).
// Here every wave writes its WTM x WTN tile (bias / RoPE / activation applied) into a private LDS image with ds_write_b32, reads it
// back as float4 of one row per lane, and does the residual loads, fp32 stores and bf3 splits on 16-byte row-contiguous pieces:
// a wave instruction then covers whole 128-byte lines.  Needs ldc % 4 == 0 and 16-byte aligned y / resid (checked by the caller
// through epilogue16_lds_ok); PIXSHUF keeps the direct form.  lds: this wave's WTM x EPI_LDS_PITCH floats.
constexpr int EPI_LDS_PITCH = 36;      // floats per LDS row for WTN = 32 (+4: 16-byte aligned rows that rotate over the banks)
constexpr int EPI_IMG_PITCH = 208;     // bytes per row of the 32-row bf3 image (13 units)
// bytes of LDS one wave needs for a WTM-row tile: the fp32 tile or the 32-row bf3 image, whichever is larger
constexpr int epi_lds_wave_bytes(int wtm) { return wtm * EPI_LDS_PITCH * 4 > 32 * EPI_IMG_PITCH ? wtm * EPI_LDS_PITCH * 4 : 32 * EPI_IMG_PITCH; }

template <int TM, int TN, bool FULL, int EPI, int O3>
__device__ __forceinline__ void gemm_epilogue16_lds_body(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][TN], int m0, int n0,
                                                         int wrow0, int wcol0, int lane, float* lds, float* amax) {
    static_assert(TN == 2, "wave tiles 32 columns wide");
    constexpr int WTM = TM * 16, WTN = TN * 16, LPR = WTN / 4, RPI = 64 / LPR, ITERS = WTM / RPI;      // lanes per row, rows per pass
    const a3r_epilogue& ep = g.epi;
    const int quad = lane >> 4, lcol = lane & 15;
    const int epi = EPI >= 0 ? EPI : ep.epi;
    const bool only3 = O3 == 0 ? false : ep.out_bf3 != 0;
    // value of element e of accumulator (i, j) after bias / RoPE / activation
    auto value = [&](int i, int j, int e, float bias, float bias_o, bool do_rope, bool rope_x, bool second) {
        float v = acc[i][j][e] + bias;
        if (do_rope) {
            const int row = m0 + wrow0 + i * 16 + quad * 4 + e;
            const float other = acc[i][j ^ 1][e] + bias_o;
            const int tok = row % ep.tokens_per_image;
            const int py = tok / ep.grid_w, px = tok - py * ep.grid_w;
            const int pp = rope_x ? px : py;
            const float c = ep.rope_cos[pp * 16 + lcol], sn = ep.rope_sin[pp * 16 + lcol];
            v = second ? v * c + other * sn : v * c - other * sn;
        }
        if (epi == A3R_EPI_GELU) v = gelu_erf(v);
        else if (epi == A3R_EPI_RELU) v = fmaxf(v, 0.f);
        else if ((epi == A3R_EPI_RESID || epi == A3R_EPI_RESID2) && ep.relu_acc) v = fmaxf(v, 0.f);
        return v;
    };
    if (O3 != 0 && only3) {
        // ---- bf3 output only: the split planes go to LDS as the final memory image of 32 rows x 32 columns (192 B per row), and
        // come back one 16-byte unit per lane, consecutive lanes = consecutive units: every store instruction writes 1 KB of
        // whole cache lines (row pairs are 384 contiguous bytes in the pair layout, rows 192 in the plain one)
        static_assert(TM % 2 == 0, "halves of 32 rows");
        constexpr int IMG_PITCH = EPI_IMG_PITCH;                     // 13 units: rows rotate over the LDS banks
        char* img = reinterpret_cast<char*>(lds);
        const int pair3 = ep.out_pair;
        char* out3 = reinterpret_cast<char*>(P.C);
        const int kcol0 = n0 + wcol0;                                // a multiple of 32: one whole k block of the output rows
#pragma unroll
        for (int half = 0; half < TM / 2; half++) {
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int colbase = n0 + wcol0 + j * 16;
                const int col = colbase + lcol;
                const bool col_ok = FULL || col < g.N;
                const float bias = (P.bias && col_ok) ? P.bias[col] : 0.f;
                const float bias_o = (epi == A3R_EPI_ROPE && P.bias) ? P.bias[min(col ^ 16, g.N - 1)] : 0.f;
                const bool do_rope = epi == A3R_EPI_ROPE && colbase < ep.rope_cols;
                const bool rope_x = (colbase & 32) != 0, second = (colbase & 16) != 0;
                const int cw = j * 16 + lcol;
                char* dcol = img + (cw >> 3) * 48 + (cw & 7) * 2;
#pragma unroll
                for (int il = 0; il < 2; il++)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int i = half * 2 + il;
                        const float v0 = value(i, j, e, bias, bias_o, do_rope, rope_x, second);
                        const float v1 = value(i, j, e + 1, bias, bias_o, do_rope, rope_x, second);
                        uint32_t p0, p1, p2;
                        bf3_split2(v0, v1, p0, p1, p2);                // low half: row e, high half: row e + 1
                        char* d = dcol + (il * 16 + quad * 4 + e) * IMG_PITCH;
                        *reinterpret_cast<uint16_t*>(d) = (uint16_t)p0;
                        *reinterpret_cast<uint16_t*>(d + 16) = (uint16_t)p1;
                        *reinterpret_cast<uint16_t*>(d + 32) = (uint16_t)p2;
                        *reinterpret_cast<uint16_t*>(d + IMG_PITCH) = (uint16_t)(p0 >> 16);
                        *reinterpret_cast<uint16_t*>(d + IMG_PITCH + 16) = (uint16_t)(p1 >> 16);
                        *reinterpret_cast<uint16_t*>(d + IMG_PITCH + 32) = (uint16_t)(p2 >> 16);
                    }
            }
            A3R_EPI_FENCE();
#pragma unroll
            for (int it = 0; it < 6; it++) {                          // 32 rows x 12 units = 6 x 64 lanes
                const int u = it * 64 + lane, r = u / 12, un = u - r * 12;
                const int grow = m0 + wrow0 + half * 32 + r;
                const u32x4 dv = *reinterpret_cast<const u32x4*>(img + r * IMG_PITCH + un * 16);
                if ((FULL || (grow < g.M && kcol0 + (un / 3) * 8 < g.N)))
                    *reinterpret_cast<u32x4*>(out3 + bf3_row_offset(grow, g.N, pair3) + bf3_k_offset(kcol0, pair3) + un * 16) = dv;
            }
            A3R_EPI_FENCE();                                          // the reads are ordered before the next half overwrites the image
        }
        return;
    }
    // ---- phase 1: bias, RoPE, activation in the accumulator layout -> LDS
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int colbase = n0 + wcol0 + j * 16;
        const int col = colbase + lcol;
        const bool col_ok = FULL || col < g.N;
        const float bias = (P.bias && col_ok) ? P.bias[col] : 0.f;
        const float bias_o = (epi == A3R_EPI_ROPE && P.bias) ? P.bias[min(col ^ 16, g.N - 1)] : 0.f;
        const bool do_rope = epi == A3R_EPI_ROPE && colbase < ep.rope_cols;
        const bool rope_x = (colbase & 32) != 0, second = (colbase & 16) != 0;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                lds[(i * 16 + quad * 4 + e) * EPI_LDS_PITCH + j * 16 + lcol] = value(i, j, e, bias, bias_o, do_rope, rope_x, second);
    }
    A3R_EPI_FENCE();                             // this wave's LDS writes are ordered before its reads (the image is wave-private)
    // ---- phase 2: one float4 of one row per lane: residuals, fp32 store, the auxiliary bf3 form
    char* out3 = O3 == 0 ? nullptr : static_cast<char*>(ep.aux_bf3);
    char* out2 = static_cast<char*>(ep.aux_fh2);              // the auxiliary output in fh2 form (fh2 kernels)
    const bool relu3 = ep.aux_relu;
    const int r_in = lane / LPR, c4 = (lane % LPR) * 4;
    const int gcol = n0 + wcol0 + c4;
    const bool col_ok = FULL || gcol < g.N;                  // N % 4 == 0 on this path: a float4 is in or out as a whole
    // residual rows first, ALL of them, then the stores: the residual may alias y (the in-place residual stream), so the compiler
    // must keep every load ahead of the later stores it could alias -- written as one loop that is a chain of ITERS dependent
    // global round trips; a lane only ever stores where it loaded, so loading everything up front is safe
    f32x4 rs[ITERS];
    if (epi == A3R_EPI_RESID || epi == A3R_EPI_RESID2) {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int grow = m0 + wrow0 + it * RPI + r_in;
            const bool ok = col_ok && (FULL || grow < g.M);
            const size_t o = (size_t)grow * g.ldc + gcol;
            rs[it] = ok ? *reinterpret_cast<const f32x4*>(P.resid + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi == A3R_EPI_RESID2 && ok) rs[it] += *reinterpret_cast<const f32x4*>(P.resid2 + o);
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int rl = it * RPI + r_in;
        const int grow = m0 + wrow0 + rl;
        f32x4 v = *reinterpret_cast<const f32x4*>(lds + rl * EPI_LDS_PITCH + c4);
        const bool ok = col_ok && (FULL || grow < g.M);
        if (!ok) continue;
        const size_t o = (size_t)grow * g.ldc + gcol;
        if (epi == A3R_EPI_RESID || epi == A3R_EPI_RESID2) {
            v += rs[it];
            if (ep.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        *reinterpret_cast<f32x4*>(P.C + o) = v;
        if (out3 || out2) {
            if (relu3) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (out3) bf3_store4(out3 + bf3_row_offset(grow, g.N, 0), gcol, v, 0);
            else {
                v = v * P.out_scale;                              // fh2 kernels only (their launch code sets out_scale >= 2^-60)
                *amax = fh2_amax4(*amax, v);
                fh2_store4(out2 + (size_t)grow * fh2_row_bytes(g.N), gcol, v);
            }
        }
    }
}

// whether the LDS form applies to this launch (wave-uniform; otherwise the direct form runs)
__device__ __forceinline__ bool epilogue16_lds_ok(const GemmArgs& g, const GroupPtrs& P) {
    const a3r_epilogue& ep = g.epi;
    if (g.direct_epilogue || ep.epi == A3R_EPI_PIXSHUF || (g.ldc & 3) || (g.N & 3)) return false;
    uintptr_t a = reinterpret_cast<uintptr_t>(P.C);
    if (ep.epi == A3R_EPI_RESID || ep.epi == A3R_EPI_RESID2) a |= reinterpret_cast<uintptr_t>(P.resid);
    if (ep.epi == A3R_EPI_RESID2) a |= reinterpret_cast<uintptr_t>(P.resid2);
    return (a & 15) == 0;
}

template <int TM, int TN, bool FULL>
__device__ __forceinline__ void gemm_epilogue16_lds(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][TN], int m0, int n0,
                                                    int wrow0, int wcol0, int lane, float* lds, float* amax) {
    const a3r_epilogue& ep = g.epi;
    const bool has3 = ep.out_bf3 || ep.aux_bf3;
#define A3R_EPI_LDS(E, O) gemm_epilogue16_lds_body<TM, TN, FULL, E, O>(g, P, acc, m0, n0, wrow0, wcol0, lane, lds, amax)
    if (!has3) {
        if (ep.epi == A3R_EPI_RESID) A3R_EPI_LDS(A3R_EPI_RESID, 0);
        else if (ep.epi == A3R_EPI_NONE) A3R_EPI_LDS(A3R_EPI_NONE, 0);
        else A3R_EPI_LDS(-1, 0);
    } else {
        if (ep.epi == A3R_EPI_ROPE) A3R_EPI_LDS(A3R_EPI_ROPE, 1);
        else if (ep.epi == A3R_EPI_GELU) A3R_EPI_LDS(A3R_EPI_GELU, 1);
        else if (ep.epi == A3R_EPI_RESID) A3R_EPI_LDS(A3R_EPI_RESID, 1);
        else if (ep.epi == A3R_EPI_RESID2) A3R_EPI_LDS(A3R_EPI_RESID2, 1);
        else if (ep.epi == A3R_EPI_RELU) A3R_EPI_LDS(A3R_EPI_RELU, 1);
        else A3R_EPI_LDS(A3R_EPI_NONE, 1);
    }
#undef A3R_EPI_LDS
}

static inline int check_epilogue(const a3r_epilogue* e, int M, int N, const char* who, bool bf3_kernel = false, bool fh2_kernel = false) {
    if (!e) return A3R_OK;
    A3R_CHECK_ARG(fh2_kernel || (!e->out_fh2 && !e->aux_fh2), "%s: out_fh2 / aux_fh2 are only available on the fh2 kernels", who);
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
    A3R_CHECK_ARG(e->epi >= A3R_EPI_NONE && e->epi <= A3R_EPI_HEAD, "%s: unknown epilogue %d", who, e->epi);
    if (e->epi == A3R_EPI_HEAD)
        A3R_CHECK_ARG(fh2_kernel && N == 128 && e->head_w && e->head_b && e->head_conf && !e->out_fh2 && !e->aux_fh2 && !e->out_bf3 && !e->aux_bf3,
                      "%s: the HEAD epilogue is available on the fh2 kernels for N == 128 with head_w, head_b, head_conf and no other output form", who);
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
