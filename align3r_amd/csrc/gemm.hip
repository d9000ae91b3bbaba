// fp32 MFMA GEMM / implicit-GEMM conv core for gfx950.
//
// Serves every dense contraction of the pair forward:
//   nn.Linear (qkv, proj, fc1, fc2, projq/k/v, decoder_embed, zero-convs)   croco/models/blocks.py:58-169
//   1x1 convs / ConvTranspose2d(k=s) of the DPT adapter                     croco/models/dpt_block.py:353-405
//   3x3 convs (stride 1/2) of the DPT head as implicit GEMM                 croco/models/dpt_block.py:33-142,323-329
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products and fp32 accumulation (a k-ordered fmaf
// chain), so results differ from the reference's fp32 CPU path only by summation order.
//
// Tiling (64-wide waves): workgroup = 4 waves = 128x128 output tile, each wave a 64x64 sub-tile held as
// 2x2 MFMA accumulators (64 VGPRs); K is walked in 32-deep slabs staged global -> registers -> LDS
// (double-buffered, rows padded 32 -> 36 floats so ds_read_b128 of 16 different rows is conflict-free).
// Inside a slab the k index is permuted so that one ds_read_b128 feeds four consecutive MFMAs:
// lane half h of MFMA step t consumes k = 8*kb + 4*h + t for both operands (sums commute).
// Workgroup ids are remapped so that the 8 XCDs each walk a contiguous run of tiles (private L2s).
#include "common.h"

namespace a3r {

constexpr int BM = 128, BN = 128, BK = 32, LDP = 36;
constexpr int GEMM_LDS_BYTES = 2 * (BM + BN) * LDP * 4;   // 73,728 B

struct GemmArgs {
    const float* A; int lda;
    const float* Wt;        // [N, K]
    float* C; int ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    a3r_epilogue epi;
    // implicit conv (AMODE 1): A = x [B, H, W, Cin]
    int cH, cW, cCin, cHo, cWo, cStride;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [2][BM][LDP]
    float* Bs = As + 2 * BM * LDP;                              // [2][BN][LDP]

    // XCD-aware bijective remap (blocks b and b+8 share an XCD)
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tile_m = wgid / g.tiles_n, tile_n = wgid - tile_m * g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    // per-thread row bookkeeping for the 4 A rows and 4 B rows it stages
    const float* a_ptr[4];
    bool a_ok[4];
    int a_oy[4], a_ox[4];
    const float* b_ptr[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = lrow + 32 * i;
        const int gm = m0 + row;
        a_ok[i] = gm < g.M;
        if (AMODE == 0) {
            a_ptr[i] = g.A + (size_t)(a_ok[i] ? gm : 0) * g.lda + lc4;
            a_oy[i] = a_ox[i] = 0;
        } else {
            const int mm = a_ok[i] ? gm : 0;
            const int hw = g.cHo * g.cWo;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
            a_oy[i] = oy * g.cStride - 1;
            a_ox[i] = ox * g.cStride - 1;
            a_ptr[i] = g.A + (size_t)b * g.cH * g.cW * g.cCin + lc4;
        }
        const int gn = n0 + row;
        b_ok[i] = gn < g.N;
        b_ptr[i] = g.Wt + (size_t)(b_ok[i] ? gn : 0) * g.K + lc4;
    }

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
        int dy = 0, dx = 0, ci0 = 0;
        if (AMODE == 1) {
            const int tap = k0 / g.cCin;
            ci0 = k0 - tap * g.cCin;
            dy = tap / 3; dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (AMODE == 0) {
                if (a_ok[i]) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
            } else {
                const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
                if (a_ok[i] && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW)
                    v = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((size_t)iy * g.cW + ix) * g.cCin + ci0);
            }
            if (g.epi.relu_a) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            ra[i] = v;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) w = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
            rb[i] = w;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = lrow + 32 * i;
            *reinterpret_cast<f32x4*>(As + (buf * BM + row) * LDP + lc4) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (buf * BN + row) * LDP + lc4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int nk = g.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* Ab = As + (buf * BM + wm * 64 + frow) * LDP + fk;
        const float* Bb = Bs + (buf * BN + wn * 64 + frow) * LDP + fk;
#pragma unroll
        for (int kb = 0; kb < BK / 8; kb++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + kb * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + 32 * LDP + kb * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bb + kb * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bb + 32 * LDP + kb * 8);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    const a3r_epilogue& ep = g.epi;
    const int half = lane >> 5, lcol = lane & 31;
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int col = n0 + wn * 64 + nt * 32 + lcol;
        const bool col_ok = col < g.N;
        const float bias = (ep.bias && col_ok) ? ep.bias[ep.epi == A3R_EPI_PIXSHUF ? col % ep.ps_cout : col] : 0.f;
        const bool do_rope = ep.epi == A3R_EPI_ROPE && (n0 + wn * 64) < ep.rope_cols;   // wave-uniform
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                float v = acc[mt][nt][e] + bias;
                if (do_rope) {
                    // pairs (d, d+16) inside each 32-wide half of the head; nt = 0 -> y position, 1 -> x
                    const float other = __shfl_xor(v, 16);
                    const int tok = row % ep.tokens_per_image;
                    const int py = tok / ep.grid_w, px = tok - py * ep.grid_w;
                    const int p = nt == 0 ? py : px;
                    const float c = ep.rope_cos[p * 16 + (lcol & 15)], s = ep.rope_sin[p * 16 + (lcol & 15)];
                    v = (lcol & 16) ? v * c + other * s : v * c - other * s;
                }
                if (row < g.M && col_ok) {
                    switch (ep.epi) {
                        case A3R_EPI_GELU: v = gelu_erf(v); break;
                        case A3R_EPI_RELU: v = fmaxf(v, 0.f); break;
                        case A3R_EPI_RESID: v = ep.resid[(size_t)row * g.ldc + col] + v; break;
                        case A3R_EPI_RESID2:
                            v = ep.resid[(size_t)row * g.ldc + col] + ep.resid2[(size_t)row * g.ldc + col] + v;
                            break;
                        default: break;
                    }
                    if (ep.epi == A3R_EPI_PIXSHUF) {
                        const int s = ep.ps_s, hw = ep.ps_h * ep.ps_w;
                        const int b = row / hw, rem = row - b * hw;
                        const int y = rem / ep.ps_w, x = rem - y * ep.ps_w;
                        const int tap = col / ep.ps_cout, co = col - tap * ep.ps_cout;
                        const int dy = tap / s, dx = tap - dy * s;
                        const size_t opix = ((size_t)b * ep.ps_h * s + (y * s + dy)) * (ep.ps_w * s) + (x * s + dx);
                        g.C[opix * ep.ps_cout + co] = v;
                    } else {
                        g.C[(size_t)row * g.ldc + col] = v;
                    }
                }
            }
        }
    }
}

static int launch_gemm(int amode, GemmArgs& g, hipStream_t st) {
    static bool attr_done[2] = {false, false};
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const int nwg = g.tiles_m * g.tiles_n;
    ProfScope prof(amode == 0 ? PK_LINEAR : PK_CONV, 2.0 * g.M * g.N * g.K, st);
    if (amode == 0) {
        if (!attr_done[0]) {
            A3R_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
            attr_done[0] = true;
        }
        hipLaunchKernelGGL(gemm_kernel<0>, dim3(nwg), dim3(256), GEMM_LDS_BYTES, st, g);
    } else {
        if (!attr_done[1]) {
            A3R_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
            attr_done[1] = true;
        }
        hipLaunchKernelGGL(gemm_kernel<1>, dim3(nwg), dim3(256), GEMM_LDS_BYTES, st, g);
    }
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

static int check_epilogue(const a3r_epilogue* e, int M, int N, const char* who) {
    if (!e) return A3R_OK;
    A3R_CHECK_ARG(e->epi >= A3R_EPI_NONE && e->epi <= A3R_EPI_PIXSHUF, "%s: unknown epilogue %d", who, e->epi);
    if (e->epi == A3R_EPI_RESID || e->epi == A3R_EPI_RESID2) A3R_CHECK_ARG(e->resid, "%s: RESID epilogue without resid", who);
    if (e->epi == A3R_EPI_RESID2) A3R_CHECK_ARG(e->resid2, "%s: RESID2 epilogue without resid2", who);
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

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_linear(const float* x, int lda, const float* w, float* y, int ldc, int M, int N, int K,
                          const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(x && w && y, "a3r_linear: null pointer");
    A3R_CHECK_ARG(M > 0 && N > 0 && K > 0, "a3r_linear: M, N, K must be positive (got %d, %d, %d)", M, N, K);
    A3R_CHECK_ARG(K % BK == 0, "a3r_linear: K (%d) must be a multiple of %d", K, BK);
    A3R_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= 1, "a3r_linear: bad leading dimensions lda=%d ldc=%d", lda, ldc);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0,
                  "a3r_linear: x and w must be 16-byte aligned");
    if (int rc = check_epilogue(epi, M, N, "a3r_linear")) return rc;
    GemmArgs g = {};
    g.A = x; g.lda = lda; g.Wt = w; g.C = y; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    if (epi) g.epi = *epi;
    if (g.epi.epi != A3R_EPI_PIXSHUF) A3R_CHECK_ARG(ldc >= N, "a3r_linear: ldc (%d) < N (%d)", ldc, N);
    return launch_gemm(0, g, as_stream(stream));
}

extern "C" int a3r_conv3x3(const float* x, const float* wp, float* y, int B, int H, int W, int Cin, int Cout,
                           int stride, const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(x && wp && y, "a3r_conv3x3: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "a3r_conv3x3: bad shape");
    A3R_CHECK_ARG(Cin > 0 && Cin % BK == 0, "a3r_conv3x3: Cin (%d) must be a multiple of %d", Cin, BK);
    A3R_CHECK_ARG(stride == 1 || stride == 2, "a3r_conv3x3: stride must be 1 or 2");
    GemmArgs g = {};
    g.cH = H; g.cW = W; g.cCin = Cin; g.cStride = stride;
    g.cHo = (H + 2 - 3) / stride + 1;
    g.cWo = (W + 2 - 3) / stride + 1;
    g.A = x; g.lda = 0; g.Wt = wp; g.C = y; g.ldc = Cout;
    g.M = B * g.cHo * g.cWo; g.N = Cout; g.K = 9 * Cin;
    if (int rc = check_epilogue(epi, g.M, g.N, "a3r_conv3x3")) return rc;
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_PIXSHUF && g.epi.epi != A3R_EPI_ROPE, "a3r_conv3x3: unsupported epilogue");
    return launch_gemm(1, g, as_stream(stream));
}
