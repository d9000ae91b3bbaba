// fp32 nn.Linear on the fp16 matrix cores of gfx950 ("fh2" GEMM): y = x W^T from two-plane fp16 splits of both operands (fh2.h),
// three v_mfma_f32_16x16x32_f16 passes per 32-deep k-step -- h0 g0 + h0 g1 + h1 g0, every fp16 x fp16 product exact in the fp32
// accumulator, 22-bit operands: as accurate against float64 as the exact-fp32 MFMA GEMM (tests/test_gpu_fh2.py), at HALF the matrix
// passes and two thirds of the operand bytes of the three-plane bf16 form (gemm_bf3.hip).  Serves the transformer call sites of
// a3r_linear (croco/models/blocks.py:58-169 qkv / proj / fc1 / fc2 / projq / projk / projv, patch embeddings, decoder_embed,
// zero-convs); the DPT convolutions and the attention products stay on the bf3 kernels.
//
// Structure (64-wide waves), as gemm_bf3.hip: a workgroup computes a BM x BN tile with WM x WN waves of 16x16 accumulator tiles;
// K is walked in 32-deep stages copied global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), NS stages deep, one raw s_barrier per
// stage with a counted s_waitcnt vmcnt.  A stage's LDS image is [rows][8 units] (128 bytes = one cache line per row); ds_read_b128
// conflicts are removed by an XOR swizzle of the unit index applied on the DMA's SOURCE side (the LDS image is lane-linear):
// physical unit j of row r holds logical unit j ^ swz(r), swz(r) = ((r >> 1) & 1) | (((r >> 3) & 1) << 2), conflict-free for the
// 16x16x32 operand pattern (rows = lane & 15, k-group = lane >> 4) under the documented ds_read_b128 lane groups.
// Epilogues: the accumulator is multiplied by the exact power of two that undoes the weight scale, then handed to the epilogues of
// gemm_common.h (bias, GELU, residuals, RoPE, fp32 / bf3 outputs); out_fh2 writes the two-plane fp16 form for the next GEMM.
#include "gemm_common.h"
#include "fh2.h"
#include <cstdlib>
#include <type_traits>

namespace a3r {

typedef __attribute__((address_space(3))) void* fh2_lptr;

template <int N> __device__ __forceinline__ void fh2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void fh2_wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: fh2_wait_vmcnt<0>(); break;   case 1: fh2_wait_vmcnt<1>(); break;   case 2: fh2_wait_vmcnt<2>(); break;
        case 3: fh2_wait_vmcnt<3>(); break;   case 4: fh2_wait_vmcnt<4>(); break;   case 5: fh2_wait_vmcnt<5>(); break;
        case 6: fh2_wait_vmcnt<6>(); break;   case 8: fh2_wait_vmcnt<8>(); break;   case 9: fh2_wait_vmcnt<9>(); break;
        case 12: fh2_wait_vmcnt<12>(); break;
        default: fh2_wait_vmcnt<0>(); break;                                // always safe
    }
}

struct Fh2Args {
    GemmArgs g;
    float inv_wscale[4];      // per group: 1 / (w_scale x_scale), times out_scale for an out_fh2 output whose epilogue is homogeneous (not GELU)
    int gm;                   // row tiles per L2 block: workgroup ids walk gm row tiles before the next column tile (1 = row-major tile order)
};

__device__ __forceinline__ int fh2_swz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 2); }
// One LDS-DMA piece through a raw buffer resource (base, 1 GB window): lane address = base + voff + soff, 16 bytes to dst + lane * 16;
// a lane with voff >= 2^30 fetches nothing and gets zeros (tools/buf_oob_lab.hip).  A plain function on purpose: called with
// template-dependent arguments from inside the kernel template, the builtin makes the HOST pass drop the kernel's stub silently.
constexpr unsigned FH2_DMA_WINDOW = 0x40000000u, FH2_DMA_OOB = 0x80000000u;
__device__ __forceinline__ void fh2_dma16(const char* base, fh2_lptr dst, unsigned voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, FH2_DMA_WINDOW, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

// ---- out_fh2 epilogue: bias / GELU / ReLU in the accumulator layout, the two fp16 planes go to a wave-private LDS image shaped like
// the final memory (32 rows x 128 bytes per 32-column block), and come back one 16-byte unit per lane: every store instruction
// writes whole 128-byte lines.  lds: EPI_FH2_WAVE_BYTES per wave.
// x / d and x % d for 0 <= x < 2^22, d > 0, inv = 1.f / d: the float quotient is off by at most one
__device__ __forceinline__ int fast_div(int x, int d, float inv) {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
__device__ __forceinline__ int fast_mod(int x, int d, float inv) { return x - fast_div(x, d, inv) * d; }

constexpr int EPI_FH2_PITCH = 144;                       // bytes per image row (128 + 16: rows rotate over the banks)
constexpr int EPI_FH2_WAVE_BYTES = 32 * EPI_FH2_PITCH;   // 4608
// EPI: the epilogue kind as a compile-time constant (straight-line bodies: a per-element runtime switch splits the basic blocks and
// keeps the compiler from packing the arithmetic of neighbouring elements)
template <int TM, bool FULL, int EPI>
__device__ __forceinline__ void fh2_epilogue_out_body(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][2], int m0, int n0, int wrow0,
                                                      int wcol0, int lane, char* img, float& amax) {
    static_assert(TM % 2 == 0, "halves of 32 rows");
    const a3r_epilogue& ep = g.epi;
    const int quad = lane >> 4, lcol = lane & 15;
    char* out = reinterpret_cast<char*>(P.C);
    const size_t pitch = fh2_row_bytes(g.N);
    const int kcol0 = n0 + wcol0;                            // a multiple of 32: one whole 128-byte k block of the output rows
    const float inv_tokens = EPI == A3R_EPI_ROPE ? 1.f / (float)ep.tokens_per_image : 0.f;
    const float inv_gw = EPI == A3R_EPI_ROPE ? 1.f / (float)ep.grid_w : 0.f;
    // the output is stored as out_scale * value (fh2.h, RANGE).  Bias, ReLU and the RoPE rotation are homogeneous: there the launch
    // code folded out_scale into the accumulator's factor and only the bias is multiplied here; GELU is not: one more product.
    const float bscale = EPI == A3R_EPI_GELU ? 1.f : P.out_scale;
#pragma unroll
    for (int half = 0; half < TM / 2; half++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int colbase = n0 + wcol0 + j * 16;
            const int col = colbase + lcol;
            const bool col_ok = FULL || col < g.N;
            const float bias = (P.bias && col_ok) ? P.bias[col] * bscale : 0.f;
            // 2-D RoPE on the leading rope_cols columns (pairs (d, d + 16) inside each 32-wide half of a head, pos_embed.py:130-157):
            // the partner column is accumulator tile j ^ 1 of the same lane
            const bool do_rope = EPI == A3R_EPI_ROPE && colbase < ep.rope_cols;          // wave-uniform (rope_cols % 64 == 0)
            const float bias_o = (do_rope && P.bias) ? P.bias[min(col ^ 16, g.N - 1)] * bscale : 0.f;
            const bool rope_x = (colbase & 32) != 0, second = (colbase & 16) != 0;
            // a lane holds one column of four rows; the fh2 form packs neighbouring COLUMNS.  Neighbouring lanes (columns c, c + 1)
            // trade half of their rows by DPP: the even lane ends up with rows e = 0, 1 of both columns, the odd lane with rows 2, 3,
            // and each stores whole 4-byte pieces [col c | col c + 1] of a plane (2-byte stores of single elements ran into LDS
            // bank conflicts: they were most of the 12 % an fh2 output cost over an fp32 one)
            const int cw = j * 16 + (lcol & ~1);
            const bool odd = lcol & 1;
            char* dcol = img + (cw >> 3) * 32 + (cw & 7) * 2;
#pragma unroll
            for (int il = 0; il < 2; il++) {
                const int i = half * 2 + il;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = acc[i][j][e] + bias;
                    if (do_rope) {
                        const int row = m0 + wrow0 + i * 16 + quad * 4 + e;
                        const float other = acc[i][j ^ 1][e] + bias_o;
                        // token position without integer division (two runtime divisions per element were most of this
                        // epilogue's cost): exact for row < 2^22 by a float quotient and one correction step each way
                        const int tok = fast_mod(row, ep.tokens_per_image, inv_tokens);
                        const int py = fast_div(tok, ep.grid_w, inv_gw), px = tok - py * ep.grid_w;
                        const int pp = rope_x ? px : py;
                        const float c = ep.rope_cos[pp * 16 + lcol], sn = ep.rope_sin[pp * 16 + lcol];
                        // one rounding order for every element (a product, then one fused multiply-add): hipcc otherwise vectorises
                        // row pairs and contracts their halves differently, i.e. a row's result depends on its parity
                        const float t = __fmul_rn(other, sn);
                        v[e] = __fmaf_rn(v[e], c, second ? t : -t);
                    }
                    if (EPI == A3R_EPI_GELU) v[e] = gelu_erf(v[e]) * P.out_scale;
                    else if (EPI == A3R_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                }
                amax = fh2_amax2(fh2_amax2(amax, v[0], v[1]), v[2], v[3]);
                // send the rows the neighbour keeps, receive its values of the rows this lane keeps
                const float g0 = dpp_xor1(odd ? v[0] : v[2]), g1 = dpp_xor1(odd ? v[1] : v[3]);
                const float l0 = odd ? g0 : v[0], r0 = odd ? v[2] : g0;       // kept row 0: (column c, column c + 1)
                const float l1 = odd ? g1 : v[1], r1 = odd ? v[3] : g1;       // kept row 1
                uint32_t a0, a1, b0, b1;
                fh2_split2(l0, r0, a0, a1);
                fh2_split2(l1, r1, b0, b1);
                char* d = dcol + (il * 16 + quad * 4 + (odd ? 2 : 0)) * EPI_FH2_PITCH;
                *reinterpret_cast<uint32_t*>(d) = a0;
                *reinterpret_cast<uint32_t*>(d + 16) = a1;
                *reinterpret_cast<uint32_t*>(d + EPI_FH2_PITCH) = b0;
                *reinterpret_cast<uint32_t*>(d + EPI_FH2_PITCH + 16) = b1;
            }
        }
        A3R_EPI_FENCE();
#pragma unroll
        for (int it = 0; it < 4; it++) {                          // 32 rows x 8 units = 4 x 64 lanes
            const int u = it * 64 + lane, r = u >> 3, un = u & 7;
            const int grow = m0 + wrow0 + half * 32 + r;
            const fh2_u32x4 dv = *reinterpret_cast<const fh2_u32x4*>(img + r * EPI_FH2_PITCH + un * 16);
            if (FULL || (grow < g.M && kcol0 + (un >> 1) * 8 < g.N))
                *reinterpret_cast<fh2_u32x4*>(out + (size_t)grow * pitch + (size_t)(kcol0 >> 3) * 32 + un * 16) = dv;
        }
        A3R_EPI_FENCE();                                          // the reads are ordered before the next half overwrites the image
    }
}

template <int TM, bool FULL>
__device__ __forceinline__ void fh2_epilogue_out(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][2], int m0, int n0, int wrow0,
                                                 int wcol0, int lane, char* img, float& amax) {
    switch (g.epi.epi) {                                           // wave-uniform
        case A3R_EPI_GELU: fh2_epilogue_out_body<TM, FULL, A3R_EPI_GELU>(g, P, acc, m0, n0, wrow0, wcol0, lane, img, amax); break;
        case A3R_EPI_RELU: fh2_epilogue_out_body<TM, FULL, A3R_EPI_RELU>(g, P, acc, m0, n0, wrow0, wcol0, lane, img, amax); break;
        case A3R_EPI_ROPE: fh2_epilogue_out_body<TM, FULL, A3R_EPI_ROPE>(g, P, acc, m0, n0, wrow0, wcol0, lane, img, amax); break;
        default: fh2_epilogue_out_body<TM, FULL, A3R_EPI_NONE>(g, P, acc, m0, n0, wrow0, wcol0, lane, img, amax); break;
    }
}

// ---- A3R_EPI_HEAD: relu(acc + bias) of a [BM, 128] tile stays in registers; the 128 -> 4 projection is reduced over a lane's two
// columns, the 16 lanes of a DPP row and the four column waves (through the idle stage ring, fixed order), then one thread per pixel
// applies the postprocess (heads/postprocess.py:37-58) and stores 16 bytes instead of the 512 the tile would have taken
template <int TM, bool FULL, int BM, int WM, int WN>
__device__ __forceinline__ void fh2_epilogue_head(const GemmArgs& g, const GroupPtrs& P, f32x4 (&acc)[TM][2], int m0, int wm, int wn, int lane,
                                                  char* smem) {
    static_assert(WN * 32 == 128, "four column waves of 32");
    const a3r_epilogue& ep = g.epi;
    const int quad = lane >> 4, lcol = lane & 15;
    float w4[4][2], bias[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = wn * 32 + j * 16 + lcol;
        bias[j] = P.bias ? P.bias[col] : 0.f;
#pragma unroll
        for (int o = 0; o < 4; o++) w4[o][j] = ep.head_w[o * 128 + col];
    }
    __syncthreads();                                            // every wave is done reading the last stage: the ring is free
    f32x4* part = reinterpret_cast<f32x4*>(smem);               // [BM rows][WN]
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float v = fmaxf(acc[i][j][e] + bias[j], 0.f);
#pragma unroll
                for (int o = 0; o < 4; o++) s[o] = __fmaf_rn(v, w4[o][j], s[o]);
            }
#pragma unroll
            for (int o = 0; o < 4; o++) s[o] = dpp_row_sum16(s[o]);
            if (lcol == 0) part[(wm * (TM * 16) + i * 16 + quad * 4 + e) * WN + wn] = f32x4{s[0], s[1], s[2], s[3]};
        }
    __syncthreads();
    const float b0 = ep.head_b[0], b1 = ep.head_b[1], b2 = ep.head_b[2], b3 = ep.head_b[3];
    for (int r = threadIdx.x; r < BM; r += WM * WN * 64) {
        const long row = (long)m0 + r;
        if (!FULL && row >= g.M) continue;
        f32x4 t = part[r * WN];
#pragma unroll
        for (int w = 1; w < WN; w++) t += part[r * WN + w];
        const float a0 = t.x + b0, a1 = t.y + b1, a2 = t.z + b2, a3 = t.w + b3;
        // reference order: xyz / d.clip(1e-8) * expm1(d)   (postprocess.py:37-46)
        const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
        const float dd = fmaxf(d, 1e-8f), em = expm1f(d);
        P.C[row * 3 + 0] = a0 / dd * em;
        P.C[row * 3 + 1] = a1 / dd * em;
        P.C[row * 3 + 2] = a2 / dd * em;
        ep.head_conf[row] = 1.f + expf(a3);
    }
}

// PASSES 3: the fp32-grade product h0 g0 + h0 g1 + h1 g0.  PASSES 1 (a3r_fh2_set_passes(1), the 16-bit operand mode): h0 g0 alone --
// plain fp16 operands (11 significant bits, inside fp16's range by the same per-site scales), fp32 accumulation; the second planes
// are neither read from LDS nor multiplied.
template <int BM, int BN, int WM, int WN, int NS, bool FULL, int AMODE = 0, int PASSES = 3>
__global__ __launch_bounds__(WM * WN * 64, (BM / WM) * (BN / WN) > 2048 ? 2 : 4) void gemm_fh2_kernel(Fh2Args fa) {
    const GemmArgs& g = fa.g;
    constexpr int NT = WM * WN * 64, U = 8;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int SA = BM * U, SB = BN * U;                                 // 16-byte units per stage
    static_assert(SA % NT == 0 && SB % NT == 0, "whole DMA rounds");
    constexpr int LA = SA / NT, LB = SB / NT, LPS = LA + LB;               // DMAs per thread per stage
    static_assert(WTN % 32 == 0 && WTM % 32 == 0, "wave tiles are multiples of 32 columns (epilogue blocks) and of 32 rows");
    constexpr int STAGE = (SA + SB) * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware bijective remap (blocks b and b+8 share an XCD), then group / tile decomposition
    const int nwg = g.tiles_per_group * g.groups;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int grp = wgid / g.tiles_per_group;
    wgid -= grp * g.tiles_per_group;
    const GroupPtrs& P = g.grp[grp];
    // L2 blocking: the workgroups resident on an XCD at one time (a run of consecutive ids) cover gm row tiles x (run / gm) column
    // tiles instead of one or two row tiles x all column tiles, so fewer distinct W tiles stream through the XCD's 4 MB L2 per A tile
    int tile_m, tile_n;
    {
        const int per = fa.gm * g.tiles_n, sm = wgid / per, rem = wgid - sm * per;
        const int rows = min(fa.gm, g.tiles_m - sm * fa.gm);
        tile_n = rem / rows;
        tile_m = sm * fa.gm + (rem - tile_n * rows);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const size_t pitch = fh2_row_bytes(g.K);
    // AMODE 0: A is an fh2 matrix [M, K].  AMODE 1: implicit 3x3 conv (padding 1, stride 1 or 2) over an fh2 channels-last map
    // x [B, H, W, Cin]: row m of A is output pixel m, its K axis is (tap, ci) -- stage kt covers 32 channels of ONE tap, i.e. one
    // 128-byte line of one input pixel, or zeros where the tap falls into the padding.
    // DMA sources as raw buffer resources (buffer_load_dwordx4 ... lds): a per-lane 32-bit byte offset relative to the tile's first row
    // and the k offset in an SGPR -- no vector address arithmetic per piece (the flat form added two 64-bit VALU operations to every
    // DMA and kept a pointer pair per piece alive).  AMODE 1: the base is moved by the tap's (uniform, possibly negative) displacement
    // and the lanes whose tap falls into the padding carry an out-of-range offset, for which the hardware writes zeros to LDS
    // (tools/buf_oob_lab.hip) and fetches nothing.
    unsigned voffA[LA], voffB[LB];
    int tapsA[LA];              // AMODE 1: bit t set <=> tap t = 3 dy + dx lies inside the map
    const char* baseA;
    if (AMODE == 0) {
        baseA = reinterpret_cast<const char*>(P.A) + (size_t)m0 * pitch;
    } else {
        const int hw = g.cHo * g.cWo;
        const int b = m0 / hw, rem = m0 - b * hw;
        const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
        baseA = reinterpret_cast<const char*>(P.A) + (((size_t)b * g.cH + oy * g.cStride) * g.cW + ox * g.cStride) * ((size_t)g.cCin * 4);
    }
#pragma unroll
    for (int i = 0; i < LA; i++) {
        const int slot = tid + NT * i, r = slot >> 3, j = slot & 7;
        const int gm = FULL ? m0 + r : min(m0 + r, g.M - 1);       // rows past M are computed on a copy of the last row, never stored
        if (AMODE == 0) {
            voffA[i] = (unsigned)(gm - m0) * (unsigned)pitch + (j ^ fh2_swz(r)) * 16;
            tapsA[i] = 0;
        } else {
            const int hw = g.cHo * g.cWo;
            const int b = gm / hw, rem = gm - b * hw;
            const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
            const int iy = oy * g.cStride, ix = ox * g.cStride;     // centre tap
            const char* centre = reinterpret_cast<const char*>(P.A) + (((size_t)b * g.cH + iy) * g.cW + ix) * ((size_t)g.cCin * 4);
            voffA[i] = (unsigned)(centre - baseA) + (j ^ fh2_swz(r)) * 16;      // pixels of one tile: monotonic in m, far below 2^30
            int mask = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int yy = iy + t / 3 - 1, xx = ix + t % 3 - 1;
                if (yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW) mask |= 1 << t;
            }
            tapsA[i] = mask;
        }
    }
#pragma unroll
    for (int i = 0; i < LB; i++) {
        const int slot = tid + NT * i, r = slot >> 3, j = slot & 7;
        const int gn = FULL ? n0 + r : min(n0 + r, g.N - 1);
        voffB[i] = (unsigned)(gn - n0) * (unsigned)pitch + (j ^ fh2_swz(r)) * 16;
    }
    const char* baseB = reinterpret_cast<const char*>(P.Wt) + (size_t)n0 * pitch;
    int c_tap = 0, c_ci = 0;                                          // AMODE 1: (tap, first channel) of the next stage to issue (stages are issued in order)
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * STAGE + wave * 1024;               // wave-uniform: the DMA adds lane * 16
        const int koff = kt * 128;
        if (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < LA; i++) fh2_dma16(baseA, (fh2_lptr)(base + NT * 16 * i), voffA[i], koff);
        } else {
            const int dy = c_tap / 3, dx = c_tap - 3 * dy;
            const long delta = ((long)(dy - 1) * g.cW + (dx - 1)) * ((long)g.cCin * 4) + (long)c_ci * 4;
#pragma unroll
            for (int i = 0; i < LA; i++)
                fh2_dma16(baseA + delta, (fh2_lptr)(base + NT * 16 * i), ((tapsA[i] >> c_tap) & 1) ? voffA[i] : FH2_DMA_OOB, 0);
            c_ci += 32;
            if (c_ci >= g.cCin) { c_ci = 0; c_tap++; }
        }
#pragma unroll
        for (int i = 0; i < LB; i++) fh2_dma16(baseB, (fh2_lptr)(base + SA * 16 + NT * 16 * i), voffB[i], koff);
    };
    const int nk = g.K / 32;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
    // lane (row = lane & 15, k-group = lane >> 4) reads, per plane p, physical unit (2 kg + p) ^ swz(row) of its row
    const int frow = lane & 15, kg = lane >> 4, sw = fh2_swz(frow);
    int offA[2], offB[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int c = (2 * kg + p) ^ sw;
        offA[p] = ((wm * WTM + frow) * U + c) * 16;
        offB[p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
    }
    // Software-pipelined form: the fragments of k-step kt+1 are read from LDS WHILE the matrix cores work on k-step kt.  The A
    // fragments roll in place (row i's registers are reloaded right after row i's MFMAs are issued), the B fragments are double
    // buffered by k-step parity.  Ring: on entry to k-step kt stage kt is in registers, stage kt+1 is waited for (own DMAs, then
    // the barrier covers everybody's), its buffer is read during the step, and the DMA of stage kt+NS goes into stage kt's
    // buffer (every wave's reads of it completed before the barrier): NS-1 stages stay in flight behind the one being read.
    const int npro = nk < NS ? nk : NS;
    for (int t = 0; t < npro; t++) issue(t, t);
    fh2_wait_vmcnt_dyn((npro - 1) * LPS);
    __builtin_amdgcn_s_barrier();
    constexpr int NPL = PASSES == 1 ? 1 : 2;                   // operand planes that are read
    f16x8 af[TM][NPL], bfr[2][TN][NPL];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int p = 0; p < NPL; p++) af[i][p] = *reinterpret_cast<const f16x8*>(smem + offA[p] + i * 16 * U * 16);
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
        for (int p = 0; p < NPL; p++) bfr[0][j][p] = *reinterpret_cast<const f16x8*>(smem + offB[p] + j * 16 * U * 16);
    int nbuf = 1 % NS;                                         // ring slot of stage kt + 1
    auto kstep = [&](int kt, f16x8 (&bc)[TN][NPL], f16x8 (&bn)[TN][NPL], auto has_next) {
        constexpr bool NEXT = decltype(has_next)::value;
        const char* sb = smem + nbuf * STAGE;
        if constexpr (NEXT) {
            __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): stage kt is in registers: its buffer may be overwritten after the barrier
            if (kt + NS <= nk) fh2_wait_vmcnt<(NS - 2) * LPS>();
            else fh2_wait_vmcnt_dyn((nk - kt - 2) * LPS);
            __builtin_amdgcn_s_barrier();
            if (kt + NS < nk) issue(kt + NS, nbuf == 0 ? NS - 1 : nbuf - 1);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int p = 0; p < NPL; p++) bn[j][p] = *reinterpret_cast<const f16x8*>(sb + offB[p] + j * 16 * U * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PASSES == 1) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bc[j][0], acc[i][j], 0, 0, 0);
                if constexpr (NEXT) {
                    __builtin_amdgcn_sched_barrier(0);
                    af[i][0] = *reinterpret_cast<const f16x8*>(sb + offA[0] + i * 16 * U * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < TM; i++) {
            // per accumulator the terms are added smallest first (x1 w0, x0 w1, x0 w0); across the row's accumulators the products of the
            // SECOND plane of A go first, so that plane's registers can be re-filled for the next stage four MFMAs earlier
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bc[j][0], acc[i][j], 0, 0, 0);
            if constexpr (NEXT) {
                __builtin_amdgcn_sched_barrier(0);
                af[i][1] = *reinterpret_cast<const f16x8*>(sb + offA[1] + i * 16 * U * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bc[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bc[j][0], acc[i][j], 0, 0, 0);
            if constexpr (NEXT) {
                __builtin_amdgcn_sched_barrier(0);
                af[i][0] = *reinterpret_cast<const f16x8*>(sb + offA[0] + i * 16 * U * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    };
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        kstep(kt, bfr[0], bfr[1], std::true_type{});
        kstep(kt + 1, bfr[1], bfr[0], std::true_type{});
    }
    if (kt + 1 < nk) {
        kstep(kt, bfr[0], bfr[1], std::true_type{});
        kstep(kt + 1, bfr[1], bfr[0], std::false_type{});
    } else {
        kstep(kt, bfr[0], bfr[1], std::false_type{});
    }
    // undo the operand scales (exact powers of two; times the output scale where the epilogue is homogeneous), then the epilogues
    const float inv = fa.inv_wscale[grp];
    float amax = 0.f;                                          // max |stored fh2 value| of this wave (range statistics, fh2.h)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = acc[i][j] * inv;
    if constexpr (BN == 128 && WN == 4 && TN == 2) {
        if (g.epi.epi == A3R_EPI_HEAD) {                       // wave-uniform
            fh2_epilogue_head<TM, FULL, BM, WM, WN>(g, P, acc, m0, wm, wn, lane, smem);
            return;
        }
    }
    // (the launch allocates max(stage ring, epilogue images) bytes of LDS)
    // the epilogues work on 32-column blocks of the wave tile
    const bool to_fh2 = g.epi.out_fh2, via_lds = !to_fh2 && epilogue16_lds_ok(g, P);       // wave-uniform
    if (to_fh2 || via_lds) __syncthreads();                    // every wave is done reading the last stage
    auto block = [&](auto cbc) {
        constexpr int cb = decltype(cbc)::value;
        f32x4 blk[TM][2];
#pragma unroll
        for (int i = 0; i < TM; i++) { blk[i][0] = acc[i][2 * cb]; blk[i][1] = acc[i][2 * cb + 1]; }
        const int wcol = wn * WTN + cb * 32;
        if (to_fh2) fh2_epilogue_out<TM, FULL>(g, P, blk, m0, n0, wm * WTM, wcol, lane, smem + wave * EPI_FH2_WAVE_BYTES, amax);
        else if (via_lds)
            gemm_epilogue16_lds<TM, 2, FULL>(g, P, blk, m0, n0, wm * WTM, wcol, lane, reinterpret_cast<float*>(smem + wave * epi_lds_wave_bytes(WTM)), &amax);
        else gemm_epilogue16<TM, 2, FULL>(g, P, blk, m0, n0, wm * WTM, wcol, lane);
    };
    block(std::integral_constant<int, 0>{});
    if constexpr (TN >= 4) block(std::integral_constant<int, 1>{});
    if constexpr (TN >= 6) block(std::integral_constant<int, 2>{});
    if constexpr (TN >= 8) block(std::integral_constant<int, 3>{});
    {
        __shared__ unsigned s_amax[WM * WN];
        fh2_publish_block(P.out_absmax, amax, s_amax);
    }
}

// fp32 [M, ldx] -> fh2 [M][K/8][2][8]: one thread per group of 8 consecutive k (32 B in, 32 contiguous bytes out)
__global__ __launch_bounds__(256) void split_fh2_kernel(const float* __restrict__ x, int ldx, char* __restrict__ y, long M, int K8, float scale,
                                                        unsigned* __restrict__ absmax) {
    const long total = M * K8;
    unsigned amax = 0;                                      // NaN-aware (fh2.h): this pass is where inputs and raw tokens enter
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / K8;
        const int kg = (int)(i - row * K8);
        const f32x4* src = reinterpret_cast<const f32x4*>(x + row * ldx + kg * 8);
        const f32x4 lo = src[0] * scale, hi = src[1] * scale;
        amax = fh2_amax_bits4(fh2_amax_bits4(amax, lo), hi);
        fh2_store8(y + row * ((size_t)K8 * 32), kg * 8, lo, hi);
    }
    __shared__ unsigned s_red[4];
    fh2_publish_block(absmax, amax, s_red);                 // every thread arrives here
}

// max |x| of n floats -> *out (a non-negative float compared as an unsigned integer; *out must be zero before the launch)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    unsigned m = 0;                                         // bit patterns with the sign cleared: a NaN weight sorts above everything
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    __shared__ unsigned s_red[4];
    fh2_publish_block(out, m, s_red);
}

struct Fh2Tile { int bm, bn, occ; double eff; };
// 0: 256x128, 16 waves (4x4), 3 stages of 48 KB, one workgroup per CU.   1: 128x64, 8 waves (4x2), two workgroups per CU, for launches
// whose tile count quantises badly.   2: 128x128, 8 waves (2x4), 2 stages of 32 KB, TWO workgroups per CU -- one workgroup's epilogue
// (and its barrier waits) run under the other's matrix work: +4 % over tile 0 on the forward's shapes although it moves 1.33x the
// operand bytes per flop (same-call A/B, tools/bench_fh2.py).   3: 256x128 with 8 waves of 64x64 (a third fewer LDS fragment reads,
// 224 VGPRs): 17 % slower (two waves per SIMD hide neither the barrier nor the longer per-wave epilogue); kept for A3R_FH2_TILE=3.
static const Fh2Tile kFh2Tiles[4] = {{256, 128, 1, 0.96}, {128, 64, 2, 0.82}, {128, 128, 2, 1.0}, {256, 128, 1, 0.0}};

static int choose_fh2_tile(int M, int N, int groups) {
    if (const char* f = getenv("A3R_FH2_TILE")) {      // developer override: 0 | 1
        const int t = atoi(f);
        if (t >= 0 && t < 4) return t;
    }
    int best_t = 0;
    double best = 1e300;
    for (int t = 0; t < 3; t++) {
        const long n = (long)((M + kFh2Tiles[t].bm - 1) / kFh2Tiles[t].bm) * ((N + kFh2Tiles[t].bn - 1) / kFh2Tiles[t].bn) * groups;
        const long slots = 256L * kFh2Tiles[t].occ;
        const double cost = (double)((n + slots - 1) / slots) * kFh2Tiles[t].bm * kFh2Tiles[t].bn * kFh2Tiles[t].occ / kFh2Tiles[t].eff;
        if (cost < best * 0.999) { best = cost; best_t = t; }
    }
    return best_t;
}

// process-wide arithmetic mode of the fh2 matrix kernels (a3r_fh2_set_passes): 3 (fp32-grade, default) or 1 (fp16 operands)
static int g_fh2_passes = 3;
int fh2_passes() { return g_fh2_passes; }

template <int BM, int BN, int WM, int WN, int NS, bool FULL, int AMODE = 0, int PASSES = 3>
static int launch_fh2_variant(const Fh2Args& fa, hipStream_t st) {
    auto kern = gemm_fh2_kernel<BM, BN, WM, WN, NS, FULL, AMODE, PASSES>;
    constexpr int ring = NS * (BM + BN) * 128, epi = WM * WN * epi_lds_wave_bytes(BM / WM), lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }));
    hipLaunchKernelGGL(kern, dim3(fa.g.tiles_per_group * fa.g.groups), dim3(WM * WN * 64), lds, st, fa);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

template <int AMODE>
static int launch_fh2(Fh2Args& fa, hipStream_t st) {
    GemmArgs& g = fa.g;
    g.direct_epilogue = 0;
    static const int gm_env = getenv("A3R_FH2_GM") ? atoi(getenv("A3R_FH2_GM")) : 0;
    int t = choose_fh2_tile(g.M, g.N, g.groups);
    if (g.epi.epi == A3R_EPI_HEAD && t != 0) t = 2;   // the HEAD epilogue is built for tiles 0 and 2
    if (AMODE == 1 && t == 3) t = 2;                  // (tile 3 is a linear-only lab tile; the 128x64 tile serves convolutions with
                                                      // 64 or 192 output channels -- the flow network's -- without a half-empty tile)
    if (g_fh2_passes == 1) t = 2;
    const int bm = kFh2Tiles[t].bm, bn = kFh2Tiles[t].bn;
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    g.tiles_per_group = g.tiles_m * g.tiles_n;
    // measured (tools/bench_fh2.py, A3R_FH2_GM=1|4|8|16): 8 row tiles per block is +6 % over the row-major order on the forward's
    // K <= 1024 shapes (fc1 +11 %) and cuts the fabric bytes by a third; the K >= 3072 shapes (2 MB of A per row tile) lose 1-2 %
    fa.gm = gm_env > 0 ? gm_env : (g.K >= 2048 ? 1 : 8);
    if (fa.gm > g.tiles_m) fa.gm = g.tiles_m;
    const bool full = g.M % bm == 0 && g.N % bn == 0;
    const double mn = (double)g.M * g.N;
    const double c_bytes = mn * (4.0 + (g.epi.aux_fh2 ? 4.0 : 0.0) + (g.epi.epi == A3R_EPI_RESID ? 4.0 : g.epi.epi == A3R_EPI_RESID2 ? 8.0 : 0.0));
    // algorithmic bytes: operands once (the conv's input map once, not once per tap), weights once, outputs once
    const double a_bytes = AMODE == 1 ? 4.0 * (g.M / (g.cHo * g.cWo)) * g.cH * g.cW * g.cCin : 4.0 * g.M * g.K;
    ProfScope prof(AMODE == 1 ? PK_CONV_FH2 : PK_LINEAR_FH2, 2.0 * g.M * g.N * g.K * g.groups, st, g.groups * (a_bytes + 4.0 * g.N * g.K + c_bytes));
    if (t == 0) {
        return full ? launch_fh2_variant<256, 128, 4, 4, 3, true, AMODE>(fa, st) : launch_fh2_variant<256, 128, 4, 4, 3, false, AMODE>(fa, st);
    }
    if (g_fh2_passes == 1)      // the 16-bit operand mode runs every shape on the 128x128 tile
        return full ? launch_fh2_variant<128, 128, 2, 4, 2, true, AMODE, 1>(fa, st) : launch_fh2_variant<128, 128, 2, 4, 2, false, AMODE, 1>(fa, st);
    if (t == 2) return full ? launch_fh2_variant<128, 128, 2, 4, 2, true, AMODE>(fa, st) : launch_fh2_variant<128, 128, 2, 4, 2, false, AMODE>(fa, st);
    if constexpr (AMODE == 0) {
        if (t == 3) return full ? launch_fh2_variant<256, 128, 4, 2, 3, true>(fa, st) : launch_fh2_variant<256, 128, 4, 2, 3, false>(fa, st);
    }
    return full ? launch_fh2_variant<128, 64, 4, 2, 3, true, AMODE>(fa, st) : launch_fh2_variant<128, 64, 4, 2, 3, false, AMODE>(fa, st);
}

// operand / output scales of one problem: 0 means 1; they must be finite and positive (powers of two by contract: only then is
// dividing them out exact)
static int fh2_scales(float x_scale, float out_scale, const char* who, float* xs, float* os) {
    *xs = x_scale == 0.f ? 1.f : x_scale;
    *os = out_scale == 0.f ? 1.f : out_scale;
    A3R_CHECK_ARG(*xs > 0.f && std::isfinite(*xs) && *os > 0.f && std::isfinite(*os), "%s: x_scale / out_scale must be positive and finite", who);
    return A3R_OK;
}
// the factor the accumulator is multiplied with: 1 / (w_scale x_scale), times out_scale when the whole epilogue of an out_fh2
// output is homogeneous of degree one (bias -- scaled in the kernel --, ReLU, RoPE); GELU is scaled after the activation, and an
// fp32 y with an aux_fh2 twin scales the twin alone
static float fh2_acc_factor(const a3r_epilogue& e, float w_scale, float xs, float os) {
    const float pre = (e.out_fh2 && e.epi != A3R_EPI_GELU) ? os : 1.f;
    return pre / (w_scale * xs);
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_fh2_set_passes(int passes) {
    const int prev = g_fh2_passes;
    if (passes == 3 || passes == 1) g_fh2_passes = passes;
    return prev;
}

extern "C" size_t a3r_fh2_bytes(long rows, int K) { return rows > 0 && K > 0 ? (size_t)rows * K * 4 : 0; }

extern "C" int a3r_split_fh2(const float* x, int ldx, void* y, long M, int K, float scale, unsigned* absmax, void* stream) {
    A3R_CHECK_ARG(x && y, "a3r_split_fh2: null pointer");
    A3R_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0, "a3r_split_fh2: K (%d) must be a positive multiple of 8 (M=%ld)", K, M);
    A3R_CHECK_ARG(ldx >= K && ldx % 4 == 0, "a3r_split_fh2: bad leading dimension %d", ldx);
    A3R_CHECK_ARG(scale > 0.f, "a3r_split_fh2: scale must be positive");
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "a3r_split_fh2: pointers must be 16-byte aligned");
    const long total = M * (K / 8);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PK_SPLIT, 8.0 * M * K, st);
    const long blocks = (total + 255) / 256;
    // persistent grid (8 workgroups per CU): one statistics atomic per workgroup, see fh2_publish_block
    hipLaunchKernelGGL(split_fh2_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, ldx,
                       static_cast<char*>(y), M, K / 8, scale, absmax);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_absmax(const float* x, long n, float* out_dev, void* stream) {
    A3R_CHECK_ARG(x && out_dev && n > 0, "a3r_absmax: bad argument");
    hipStream_t st = as_stream(stream);
    A3R_HIP(hipMemsetAsync(out_dev, 0, 4, st));
    const long blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(out_dev));
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" float a3r_fh2_weight_scale(float absmax) {
    // the power of two that puts max|w| into [2^12, 2^13): far from fp16's 65504 and with every weight above 2^-16 max|w| at full 22 bits
    if (!(absmax > 0.f) || !std::isfinite(absmax)) return 1.f;
    int e;
    std::frexp(absmax, &e);                 // absmax = f 2^e, f in [0.5, 1)
    return std::ldexp(1.f, 13 - e);
}

extern "C" int a3r_linear_fh2_grouped(const a3r_group_ptrs_fh2* groups, int n_groups, int ldc, int M, int N, int K,
                                      const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(groups && n_groups >= 1 && n_groups <= 4, "a3r_linear_fh2_grouped: 1..4 groups required");
    A3R_CHECK_ARG(M > 0 && N > 0 && K > 0, "a3r_linear_fh2: M, N, K must be positive (got %d, %d, %d)", M, N, K);
    A3R_CHECK_ARG(K % 32 == 0, "a3r_linear_fh2: K (%d) must be a multiple of 32", K);
    A3R_CHECK_ARG(ldc >= 1, "a3r_linear_fh2: bad leading dimension ldc=%d", ldc);
    if (int rc = check_epilogue(epi, M, N, "a3r_linear_fh2", true, true)) return rc;
    Fh2Args fa = {};
    GemmArgs& g = fa.g;
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(!g.epi.relu_a && !g.epi.x_pair && !g.epi.out_pair, "a3r_linear_fh2: relu_a / x_pair / out_pair do not apply to fh2 operands");
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_PIXSHUF, "a3r_linear_fh2: PIXSHUF is not available");
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_ROPE || M < (1 << 22), "a3r_linear_fh2: the ROPE epilogue takes at most 2^22 rows (got %d)", M);
    if (g.epi.aux_fh2)
        A3R_CHECK_ARG(!g.epi.out_fh2 && !g.epi.out_bf3 && N % 8 == 0 && ldc == N && g.epi.epi != A3R_EPI_PIXSHUF &&
                          (reinterpret_cast<uintptr_t>(g.epi.aux_fh2) & 15) == 0,
                      "a3r_linear_fh2: aux_fh2 needs an fp32 y with ldc == N, N %% 8 == 0 and a 16-byte aligned buffer");
    if (g.epi.out_fh2) {
        A3R_CHECK_ARG(!g.epi.out_bf3 && !g.epi.aux_bf3, "a3r_linear_fh2: out_fh2 excludes out_bf3 / aux_bf3");
        A3R_CHECK_ARG(N % 32 == 0 && ldc == N && (g.epi.epi == A3R_EPI_NONE || g.epi.epi == A3R_EPI_GELU || g.epi.epi == A3R_EPI_RELU ||
                                                  g.epi.epi == A3R_EPI_ROPE),
                      "a3r_linear_fh2: out_fh2 needs N %% 32 == 0, ldc == N and a NONE / GELU / RELU / ROPE epilogue");
    }
    for (int i = 0; i < n_groups; i++) {
        float xs, os;
        if (int rc = fh2_scales(groups[i].x_scale, groups[i].out_scale, "a3r_linear_fh2", &xs, &os)) return rc;
        g.grp[i] = {static_cast<const float*>(groups[i].x2), static_cast<const float*>(groups[i].w2), groups[i].y, groups[i].bias,
                    groups[i].resid, groups[i].resid2, os, groups[i].out_absmax};
        if (int rc = check_group(g.grp[i], g.epi.epi, "a3r_linear_fh2")) return rc;
        A3R_CHECK_ARG(groups[i].w_scale > 0.f, "a3r_linear_fh2: w_scale must be positive");
        fa.inv_wscale[i] = fh2_acc_factor(g.epi, groups[i].w_scale, xs, os);
    }
    g.groups = n_groups;
    g.lda = K; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    A3R_CHECK_ARG(ldc >= N, "a3r_linear_fh2: ldc (%d) < N (%d)", ldc, N);
    if (g.epi.out_bf3) A3R_CHECK_ARG(ldc == N, "a3r_linear_fh2: out_bf3 needs ldc == N");
    return launch_fh2<0>(fa, as_stream(stream));
}

extern "C" int a3r_linear_fh2(const void* x2, const void* w2, float w_scale, float* y, int ldc, int M, int N, int K,
                              const a3r_epilogue* epi, void* stream) {
    a3r_group_ptrs_fh2 p = {x2, w2, y, epi ? epi->bias : nullptr, epi ? epi->resid : nullptr, epi ? epi->resid2 : nullptr, w_scale,
                            epi ? epi->x_scale : 0.f, epi ? epi->out_scale : 0.f, epi ? epi->out_absmax : nullptr};
    return a3r_linear_fh2_grouped(&p, 1, ldc, M, N, K, epi, stream);
}

extern "C" int a3r_conv3x3_fh2(const void* x2, const void* wp2, float w_scale, float* y, int B, int H, int W, int Cin, int Cout, int stride,
                               const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(x2 && wp2 && y, "a3r_conv3x3_fh2: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "a3r_conv3x3_fh2: bad shape");
    A3R_CHECK_ARG(stride == 1 || stride == 2, "a3r_conv3x3_fh2: stride must be 1 or 2");
    A3R_CHECK_ARG(Cin % 32 == 0, "a3r_conv3x3_fh2: Cin (%d) must be a multiple of 32", Cin);
    A3R_CHECK_ARG(w_scale > 0.f, "a3r_conv3x3_fh2: w_scale must be positive");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long M = (long)B * Ho * Wo;
    A3R_CHECK_ARG(M < (1L << 31) && (long)B * H * W * Cin * 4 < (1L << 46), "a3r_conv3x3_fh2: map too large");
    if (int rc = check_epilogue(epi, (int)M, Cout, "a3r_conv3x3_fh2", true, true)) return rc;
    Fh2Args fa = {};
    GemmArgs& g = fa.g;
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(!g.epi.relu_a && !g.epi.x_pair && !g.epi.out_pair && !g.epi.out_bf3 && !g.epi.aux_bf3,
                  "a3r_conv3x3_fh2: relu_a / row-pair / bf3 outputs do not apply (the producer writes the pre-activated fh2 input: aux_fh2 + aux_relu)");
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_PIXSHUF && g.epi.epi != A3R_EPI_ROPE, "a3r_conv3x3_fh2: PIXSHUF / ROPE are not available");
    if (g.epi.out_fh2)
        A3R_CHECK_ARG(!g.epi.aux_fh2 && Cout % 32 == 0 && (g.epi.epi == A3R_EPI_NONE || g.epi.epi == A3R_EPI_GELU || g.epi.epi == A3R_EPI_RELU),
                      "a3r_conv3x3_fh2: out_fh2 needs Cout %% 32 == 0, a NONE / GELU / RELU epilogue and no aux_fh2");
    if (g.epi.aux_fh2)
        A3R_CHECK_ARG(Cout % 8 == 0 && (reinterpret_cast<uintptr_t>(g.epi.aux_fh2) & 15) == 0, "a3r_conv3x3_fh2: aux_fh2 needs Cout %% 8 == 0 and a 16-byte aligned buffer");
    float xs, os;
    if (int rc = fh2_scales(g.epi.x_scale, g.epi.out_scale, "a3r_conv3x3_fh2", &xs, &os)) return rc;
    g.grp[0] = {static_cast<const float*>(x2), static_cast<const float*>(wp2), y, g.epi.bias, g.epi.resid, g.epi.resid2, os, g.epi.out_absmax};
    if (int rc = check_group(g.grp[0], g.epi.epi, "a3r_conv3x3_fh2")) return rc;
    fa.inv_wscale[0] = fh2_acc_factor(g.epi, w_scale, xs, os);
    g.groups = 1;
    g.M = (int)M; g.N = Cout; g.K = 9 * Cin; g.lda = g.K; g.ldc = Cout;
    g.cH = H; g.cW = W; g.cCin = Cin; g.cHo = Ho; g.cWo = Wo; g.cStride = stride;
    return launch_fh2<1>(fa, as_stream(stream));
}
