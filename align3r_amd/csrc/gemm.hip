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
// Tiling (64-wide waves): a workgroup is 4 waves (2x2) over a BM x BN output tile, BM x BN in
// {128x128, 128x64, 64x64}; each wave holds (BM/64) x (BN/64) MFMA accumulators of 32x32.  The launcher
// picks the tile so that the grid fills the 256 CUs (the ViT-L problems are 144..1536 tiles of 128x128:
// tail quantisation, not the main loop, is what costs throughput -- see tools/gemm_lab.hip).
// K is walked in 32-deep slabs staged global -> registers -> LDS (double-buffered, rows padded 32 -> 36
// floats so ds_read_b128 of 16 different rows is conflict-free).  Inside a slab the k index is permuted so
// that one ds_read_b128 feeds four consecutive MFMAs: lane half h of MFMA step t consumes
// k = 8*kb + 4*h + t for both operands (sums commute).  Workgroup ids are remapped so that the 8 XCDs
// each walk a contiguous run of tiles (private L2s).  Up to 4 same-shape problems (different operands:
// the two decoders) can share one launch ("grouped").
#include "gemm_common.h"
#include <cstdlib>

namespace a3r {

constexpr int BK = 32, LDP = 36;

// AMODE 0: A is [M, lda] row-major.  AMODE 1: implicit 3x3 conv gather.
// FULL: M % BM == 0 && N % BN == 0 (no bounds predication; AMODE 0 only).  RELU_A: relu on the A operand.
template <int AMODE, int BM, int BN, bool FULL, bool RELU_A>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;        // MFMA tiles per wave (waves are 2 x 2)
    constexpr int WTM = BM / 2, WTN = BN / 2;        // wave tile
    constexpr int RA = BM / 32, RB = BN / 32;        // float4 per thread per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [2][BM][LDP]
    float* Bs = As + 2 * BM * LDP;                              // [2][BN][LDP]

    // XCD-aware bijective remap (blocks b and b+8 share an XCD)
    const int nwg = g.tiles_per_group * g.groups;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int grp = wgid / g.tiles_per_group;
    wgid -= grp * g.tiles_per_group;
    const GroupPtrs& P = g.grp[grp];
    const int tile_m = wgid / g.tiles_n, tile_n = wgid - tile_m * g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    const float* a_ptr[RA];
    bool a_ok[RA];
    int a_oy[RA], a_ox[RA];
    const float* b_ptr[RB];
    bool b_ok[RB];
#pragma unroll
    for (int i = 0; i < RA; i++) {
        const int gm = m0 + lrow + 32 * i;
        a_ok[i] = FULL || gm < g.M;
        if (AMODE == 0) {
            a_ptr[i] = P.A + (size_t)(a_ok[i] ? gm : 0) * g.lda + lc4;
            a_oy[i] = a_ox[i] = 0;
        } else {
            const int mm = a_ok[i] ? gm : 0;
            const int hw = g.cHo * g.cWo;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
            a_oy[i] = oy * g.cStride - 1;
            a_ox[i] = ox * g.cStride - 1;
            a_ptr[i] = P.A + (size_t)b * g.cH * g.cW * g.cCin + lc4;
        }
    }
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const int gn = n0 + lrow + 32 * i;
        b_ok[i] = FULL || gn < g.N;
        b_ptr[i] = P.Wt + (size_t)(b_ok[i] ? gn : 0) * g.K + lc4;
    }

    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int k0) {
        int dy = 0, dx = 0, ci0 = 0;
        if (AMODE == 1) {
            const int tap = k0 / g.cCin;
            ci0 = k0 - tap * g.cCin;
            dy = tap / 3; dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < RA; i++) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (AMODE == 0) {
                if (FULL || a_ok[i]) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
            } else {
                const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
                if (a_ok[i] && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW)
                    v = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((size_t)iy * g.cW + ix) * g.cCin + ci0);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (FULL || b_ok[i]) w = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
            rb[i] = w;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; i++) {
            f32x4 v = ra[i];
            if (RELU_A) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<f32x4*>(As + (buf * BM + lrow + 32 * i) * LDP + lc4) = v;
        }
#pragma unroll
        for (int i = 0; i < RB; i++) *reinterpret_cast<f32x4*>(Bs + (buf * BN + lrow + 32 * i) * LDP + lc4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
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
        const float* Ab = As + (buf * BM + wm * WTM + frow) * LDP + fk;
        const float* Bb = Bs + (buf * BN + wn * WTN + frow) * LDP + fk;
#pragma unroll
        for (int kb = 0; kb < BK / 8; kb++) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDP + kb * 8);
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDP + kb * 8);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<TM, TN, FULL>(g, P, acc, m0, n0, wm * WTM, wn * WTN, lane);
}

template <int AMODE, int BM, int BN, bool FULL, bool RELU_A>
static int launch_variant(const GemmArgs& g, hipStream_t st) {
    auto kern = gemm_kernel<AMODE, BM, BN, FULL, RELU_A>;
    constexpr int lds = 2 * (BM + BN) * LDP * 4;
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }));
    hipLaunchKernelGGL(kern, dim3(g.tiles_per_group * g.groups), dim3(256), lds, st, g);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

// Tile choice.  Workgroups resident on a CU share its MFMA pipes, so a launch of n equal workgroups takes about
// ceil(n / 256) workgroup-times whatever the residency; the cost of a tile shape is therefore
// ceil(n / 256) * BM * BN / eff, with the main-loop efficiencies measured on MI355X (tools/gemm_lab.hip:
// 131 / 119 / 113 TFLOP/s at 4096^3 for 128x128 / 128x64 / 64x64).  The model reproduces the measured winner
// on every ViT-L shape of the pair forward.
static void choose_tile(int M, int N, int groups, int* bm, int* bn) {
    static const int tiles[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    static const double eff[3] = {1.0, 0.91, 0.86};
    double best = 1e300;
    for (int t = 0; t < 3; t++) {
        const long n = (long)((M + tiles[t][0] - 1) / tiles[t][0]) * ((N + tiles[t][1] - 1) / tiles[t][1]) * groups;
        const double cost = (double)((n + 255) / 256) * tiles[t][0] * tiles[t][1] / eff[t];
        if (cost < best * 0.999) { best = cost; *bm = tiles[t][0]; *bn = tiles[t][1]; }
    }
}

static int launch_gemm(int amode, GemmArgs& g, hipStream_t st) {
    int bm, bn;
    choose_tile(g.M, g.N, g.groups, &bm, &bn);
    if (const char* f = getenv("A3R_GEMM_TILE")) {      // developer override: "128x128" | "128x64" | "64x64"
        int a = 0, b = 0;
        if (sscanf(f, "%dx%d", &a, &b) == 2 && ((a == 128 && (b == 128 || b == 64)) || (a == 64 && b == 64))) { bm = a; bn = b; }
    }
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    g.tiles_per_group = g.tiles_m * g.tiles_n;
    const bool full = g.M % bm == 0 && g.N % bn == 0;
    const bool relu = g.epi.relu_a != 0;
    ProfScope prof(amode == 0 ? PK_LINEAR : PK_CONV, 2.0 * g.M * g.N * g.K * g.groups, st);
#define A3R_TILE_DISPATCH(AM, FULLV, RELUV)                                                  \
    do {                                                                                    \
        if (bm == 128 && bn == 128) return launch_variant<AM, 128, 128, FULLV, RELUV>(g, st); \
        if (bm == 128 && bn == 64) return launch_variant<AM, 128, 64, FULLV, RELUV>(g, st);   \
        return launch_variant<AM, 64, 64, FULLV, RELUV>(g, st);                               \
    } while (0)
    if (amode == 0) {
        if (full) A3R_TILE_DISPATCH(0, true, false);
        A3R_TILE_DISPATCH(0, false, false);
    } else {
        if (relu) A3R_TILE_DISPATCH(1, false, true);
        A3R_TILE_DISPATCH(1, false, false);
    }
#undef A3R_TILE_DISPATCH
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_linear_grouped(const a3r_group_ptrs* groups, int n_groups, int lda, int ldc, int M, int N, int K,
                                  const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(groups && n_groups >= 1 && n_groups <= 4, "a3r_linear_grouped: 1..4 groups required");
    A3R_CHECK_ARG(M > 0 && N > 0 && K > 0, "a3r_linear: M, N, K must be positive (got %d, %d, %d)", M, N, K);
    A3R_CHECK_ARG(K % BK == 0, "a3r_linear: K (%d) must be a multiple of %d", K, BK);
    A3R_CHECK_ARG(lda >= K && lda % 4 == 0 && ldc >= 1, "a3r_linear: bad leading dimensions lda=%d ldc=%d", lda, ldc);
    if (int rc = check_epilogue(epi, M, N, "a3r_linear")) return rc;
    GemmArgs g = {};
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(!g.epi.relu_a, "a3r_linear: relu_a is only available on a3r_conv3x3");
    for (int i = 0; i < n_groups; i++) {
        g.grp[i] = {groups[i].x, groups[i].w, groups[i].y, groups[i].bias, groups[i].resid, groups[i].resid2};
        if (int rc = check_group(g.grp[i], g.epi.epi, "a3r_linear")) return rc;
    }
    g.groups = n_groups;
    g.lda = lda; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    if (g.epi.epi != A3R_EPI_PIXSHUF) A3R_CHECK_ARG(ldc >= N, "a3r_linear: ldc (%d) < N (%d)", ldc, N);
    return launch_gemm(0, g, as_stream(stream));
}

extern "C" int a3r_linear(const float* x, int lda, const float* w, float* y, int ldc, int M, int N, int K,
                          const a3r_epilogue* epi, void* stream) {
    a3r_group_ptrs p = {x, w, y, epi ? epi->bias : nullptr, epi ? epi->resid : nullptr, epi ? epi->resid2 : nullptr};
    return a3r_linear_grouped(&p, 1, lda, ldc, M, N, K, epi, stream);
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
    g.lda = 0; g.ldc = Cout;
    g.M = B * g.cHo * g.cWo; g.N = Cout; g.K = 9 * Cin;
    if (int rc = check_epilogue(epi, g.M, g.N, "a3r_conv3x3")) return rc;
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_PIXSHUF && g.epi.epi != A3R_EPI_ROPE, "a3r_conv3x3: unsupported epilogue");
    g.groups = 1;
    g.grp[0] = {x, wp, y, g.epi.bias, g.epi.resid, g.epi.resid2};
    if (int rc = check_group(g.grp[0], g.epi.epi, "a3r_conv3x3")) return rc;
    return launch_gemm(1, g, as_stream(stream));
}
