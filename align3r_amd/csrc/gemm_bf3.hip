// fp32 nn.Linear on the bf16 matrix cores of gfx950 ("bf3" GEMM): y = x W^T with fp32-level accuracy at ~1.5x the
// throughput of the exact-fp32 MFMA kernel in gemm.hip (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate).
//
// Both operands arrive pre-split into three bf16 planes (bf3.h: x = x0 + x1 + x2 EXACTLY), and
//     a b  =  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)  +  [a1b2 + a2b1 + a2b2, dropped: <= 2^-23 |a b|]
// is evaluated as six bf16 MFMA passes per k-step (v_mfma_f32_16x16x32_bf16, or 32x32x16).  Every bf16 x bf16 product is exact in the
// fp32 accumulator, so the only errors are the dropped terms (below one fp32 ulp of the product) and the fp32
// summation itself -- the same error class as the reference's fp32 GEMM (measured in tests/test_gpu_ops.py against
// float64: not larger than the exact-fp32 MFMA kernel's).  Serves the same call sites as a3r_linear
// (croco/models/blocks.py:58-169 qkv / proj / fc1 / fc2 / projq / projk / projv, patch embeddings, decoder_embed).
//
// Structure (64-wide waves): a workgroup computes a BM x BN tile with WM x WN waves, each wave a grid of 32x32 MFMA
// accumulators.  K is walked in BK-deep stages copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, no ds_write), NS stages deep, one raw s_barrier per stage with a counted s_waitcnt vmcnt so that the
// next stage's DMA stays in flight across the barrier.  The LDS image of a stage is the plain row-major bf3 tile
// (rows of U = 3 BK/8 16-byte units, no padding -- LDS-DMA writes lane-linear); bank conflicts are avoided by
// rotating each row's units on the SOURCE side: unit c of row r is stored at unit (c - rot(r)) mod U with
// rot(r) = (r / (16/G)) % G, G = BK/8, which makes every ds_read_b128 of the MFMA operand conflict-free
// (SQ_LDS_BANK_CONFLICT = 0 measured).  Workgroup ids are remapped so each XCD walks a contiguous run of tiles.
#include "gemm_common.h"
#include "bf3.h"
#include <cstdlib>

namespace a3r {

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt takes an immediate: dispatch over the counts that occur (stages in flight x DMAs per stage per wave)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;   case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;   case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 12: wait_vmcnt<12>(); break;
        default: wait_vmcnt<0>(); break;                                    // always safe
    }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MF: MFMA shape -- 32: v_mfma_f32_32x32x16_bf16 (two 16-deep k-steps per BK = 32 stage); 16: v_mfma_f32_16x16x32_bf16
// (one 32-deep k-step per stage; at the clocks the chip holds under bf16 MFMA load it delivers ~1.15x the FLOP/s of the
// 32x32 shape, MI355X_MICROARCH.md "DVFS give-back" (7), and measured here: tools/gemm_bf3_lab.hip).
// all-zero source for the padding taps of the implicit conv (LDS-DMA has no predicated zero fill)
__device__ __attribute__((aligned(16))) unsigned int g_bf3_zero[4];

// AMODE 0: A is a bf3 matrix [M, K].  AMODE 1: implicit 3x3 conv (padding 1, stride 1 or 2) over a bf3 channels-last map
// x [B, H, W, Cin]: row m of A is output pixel m, its K axis is (tap, ci) -- stage kt covers BK channels of ONE tap, i.e.
// 6 BK contiguous bytes of one input pixel, or zeros where the tap falls into the padding.
// NP: plane products evaluated -- 6: fp32-accurate (default); 3: a0b0 + a0b1 + a1b0 (operands effectively 16 bits, error
// ~2^-17 per product); 1: a0b0 only = plain bf16 operands with fp32 accumulation (BASELINE config 5's "bf16-MFMA mode").
template <int AMODE, int BM, int BN, int BK, int WM, int WN, int NS, int MF, bool FULL, int NP>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf3_kernel(GemmArgs g) {
    constexpr bool M16 = MF == 16;
    static_assert(MF == 32 || (MF == 16 && BK == 32), "16x16x32 MFMA needs BK = 32");
    static_assert(BK == 32, "the row-pair weight layout is addressed in 32-deep k blocks");
    constexpr int NT = WM * WN * 64, KG = BK / 8, U = 3 * KG, G = KG, PER = 16 / G;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int SA = BM * U, SB = BN * U;                                 // 16-byte units per stage
    constexpr int LA = (SA + NT - 1) / NT, LB = (SB + NT - 1) / NT;         // DMAs per thread per stage (the last one may cover only the first waves)
    static_assert(SA % 64 == 0 && SB % 64 == 0, "whole waves");
    static_assert(WTN % 32 == 0 && WTM % 32 == 0, "wave tiles are multiples of 32 (RoPE epilogue pairs columns d, d+16)");
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
    const int tile_m = wgid / g.tiles_n, tile_n = wgid - tile_m * g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const size_t pitch = (size_t)g.K * 6;
    // Source-side rotation of a row's units (the LDS image itself is lane-linear).  32x32x16 operands: 16-lane read groups
    // hold 16 different rows at one unit -> rotate single units by (r / PER) % G.  16x16x32 operands: a read group holds
    // rows {0-3, 12-15} at k-group kg and rows {4-11} at kg + 1 -> rotate whole k-groups (3 units) by 2 for rows 8..15 mod 16.
    auto src_unit = [&](int r, int cp) { return M16 ? (cp + 6 * ((r >> 3) & 1)) % U : (cp + (r / PER) % G) % U; };
    const char* srcA[LA];       // AMODE 0: row pointer at k = 0; AMODE 1: the centre tap's pixel, channel 0
    int tapsA[LA];              // AMODE 1: bit t set <=> tap t = 3 dy + dx lies inside the map
    const char* srcB[LB];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        const int slot = tid + NT * i, r = (slot / U) % BM, cp = slot % U;
        const int gm = FULL ? m0 + r : min(m0 + r, g.M - 1);       // rows past M are computed on a copy of the last row, never stored
        if (AMODE == 0) {
            srcA[i] = reinterpret_cast<const char*>(P.A) + bf3_row_offset(gm, g.K, g.epi.x_pair) + src_unit(r, cp) * 16;
            tapsA[i] = 0;
        } else {
            const int hw = g.cHo * g.cWo;
            const int b = gm / hw, rem = gm - b * hw;
            const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
            const int iy = oy * g.cStride, ix = ox * g.cStride;     // centre tap
            srcA[i] = reinterpret_cast<const char*>(P.A) + (((size_t)b * g.cH + iy) * g.cW + ix) * ((size_t)g.cCin * 6) + src_unit(r, cp) * 16;
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
        const int slot = tid + NT * i, r = (slot / U) % BN, cp = slot % U;
        const int gn = FULL ? n0 + r : min(n0 + r, g.N - 1);
        srcB[i] = reinterpret_cast<const char*>(P.Wt) + bf3_w_row_offset(gn, g.K) + src_unit(r, cp) * 16;     // row-pair weight layout (bf3.h)
    }
    const bool lastA = (LA - 1) * NT + wave * 64 < SA, lastB = (LB - 1) * NT + wave * 64 < SB;   // wave-uniform
    const int lps = LA + LB - ((SA % NT != 0 && !lastA) ? 1 : 0) - ((SB % NT != 0 && !lastB) ? 1 : 0);
    int c_tap = 0, c_ci = 0;                                          // AMODE 1: (tap, first channel) of the next stage to issue (stages are issued in order)
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * STAGE + wave * 1024;               // wave-uniform: the DMA adds lane * 16
        const size_t koff = (size_t)kt * (KG * 48) * (AMODE == 0 && g.epi.x_pair ? 2 : 1);
        long delta = 0;
        if (AMODE == 1) {
            const int dy = c_tap / 3, dx = c_tap - 3 * dy;
            delta = ((long)(dy - 1) * g.cW + (dx - 1)) * ((long)g.cCin * 6) + (long)c_ci * 6;
        }
#pragma unroll
        for (int i = 0; i < LA; i++)
            if (i + 1 < LA || SA % NT == 0 || lastA) {
                const char* src = AMODE == 0 ? srcA[i] + koff
                                             : ((tapsA[i] >> c_tap) & 1 ? srcA[i] + delta : reinterpret_cast<const char*>(g_bf3_zero));
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + NT * 16 * i), 16, 0, 0);
            }
        if (AMODE == 1) {
            c_ci += BK;
            if (c_ci >= g.cCin) { c_ci = 0; c_tap++; }
        }
#pragma unroll
        for (int i = 0; i < LB; i++)
            if (i + 1 < LB || SB % NT == 0 || lastB)
                __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + (size_t)kt * BF3_W_KBLOCK_BYTES), (lptr_t)(base + SA * 16 + NT * 16 * i), 16, 0, 0);
    };
    // Stage kt is consumed after (a) this wave's DMAs for it have landed -- a counted vmcnt that leaves the younger stages in
    // flight -- and (b) the barrier, which extends (a) to the workgroup and also says every wave is done with stage kt-1,
    // whose buffer the next issue overwrites.
    const int nk = g.K / BK;
    auto acquire = [&](int kt) {
        const int rem = nk - 1 - kt, fly = rem < NS - 2 ? rem : NS - 2;
        wait_vmcnt_dyn(fly * lps);
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
    };
    for (int t = 0; t < NS - 1 && t < nk; t++) issue(t, t);

    if constexpr (!M16) {
        constexpr int TM = WTM / 32, TN = WTN / 32;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        // lane (row = lane & 31, k-half = lane >> 5) reads, per k-step s and plane p, unit (2 s + h) 3 + p of its row
        const int frow = lane & 31, fh = lane >> 5, rot = (frow / PER) % G;
        int offA[BK / 16][3], offB[BK / 16][3];
#pragma unroll
        for (int s = 0; s < BK / 16; s++)
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const int c = ((2 * s + fh) * 3 + p - rot + U) % U;
                offA[s][p] = ((wm * WTM + frow) * U + c) * 16;
                offB[s][p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
            }
        for (int kt = 0; kt < nk; kt++) {
            acquire(kt);
            const char* sb = smem + (kt % NS) * STAGE;
#pragma unroll
            for (int s = 0; s < BK / 16; s++) {
                bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int p = 0; p < 3; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(sb + offA[s][p] + i * 32 * U * 16);
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int p = 0; p < 3; p++) bf[j][p] = *reinterpret_cast<const bf16x8*>(sb + offB[s][p] + j * 32 * U * 16);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        // smallest terms first
                        if (NP == 6) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        }
                        if (NP >= 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        }
        gemm_epilogue<TM, TN, FULL>(g, P, acc, m0, n0, wm * WTM, wn * WTN, lane);
    } else {
        constexpr int TM = WTM / 16, TN = WTN / 16;
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
        // lane (row = lane & 15, k-group = lane >> 4) reads, per plane p, unit 3 ((kg - rot) mod 4) + p of its row
        const int frow = lane & 15, kg = lane >> 4, rotk = 2 * ((frow >> 3) & 1);
        int offA[3], offB[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int c = 3 * ((kg - rotk + 4) % 4) + p;
            offA[p] = ((wm * WTM + frow) * U + c) * 16;
            offB[p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
        }
        for (int kt = 0; kt < nk; kt++) {
            acquire(kt);
            const char* sb = smem + (kt % NS) * STAGE;
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int p = 0; p < 3; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(sb + offA[p] + i * 16 * U * 16);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int p = 0; p < 3; p++) bf[j][p] = *reinterpret_cast<const bf16x8*>(sb + offB[p] + j * 16 * U * 16);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (NP == 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                    }
                    if (NP >= 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        }
        static_assert(NS * STAGE >= WM * WN * epi_lds_wave_bytes(WTM) || TN != 2, "the LDS epilogue image fits in the stage ring");
        if (TN == 2 && epilogue16_lds_ok(g, P)) {                 // wave-uniform
            __syncthreads();                                       // every wave is done reading the last stage
            if constexpr (TN == 2)
            {
                float amax_unused = 0.f;                           // (range statistics exist on the fh2 kernels only)
                gemm_epilogue16_lds<TM, TN, FULL>(g, P, acc, m0, n0, wm * WTM, wn * WTN, lane,
                                                  reinterpret_cast<float*>(smem + wave * epi_lds_wave_bytes(WTM)), &amax_unused);
            }
        } else {
            gemm_epilogue16<TM, TN, FULL>(g, P, acc, m0, n0, wm * WTM, wn * WTN, lane);
        }
    }
}

// ---- 256 x 256 tile ("W in two halves"), nn.Linear only.  Operand delivery L2 -> LDS, not the matrix cores, is what bounds
// the kernel above, so this variant spends the same 144 KB of LDS on a tile with 2/3 of the delivery bytes per
// flop: A stages are double-buffered (2 x 48 KB), W holds ONE stage as two 128-column halves (2 x 24 KB).  Eight waves, each a
// 32-row strip over all 256 columns (2 x 16 accumulator tiles): a stage multiplies the strip by half 0, then by half 1; half 0
// is refilled for the next stage at the mid-stage barrier, half 1 and the next A stage at the stage barrier.
template <bool FULL, int NP>
__global__ __launch_bounds__(512) void gemm_bf3_w2h_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, HN = 128, NT = 512, U = 12;
    constexpr int A_BYTES = BM * U * 16, H_BYTES = HN * U * 16;
    constexpr int LA = BM * U / NT, LH = HN * U / NT;                      // 6, 3 DMAs per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [2][256 rows][12 units]
    char* Hs = smem + 2 * A_BYTES;          // [2 halves][128 rows][12 units]

    const int nwg = g.tiles_per_group * g.groups;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int grp = wgid / g.tiles_per_group;
    wgid -= grp * g.tiles_per_group;
    const GroupPtrs& P = g.grp[grp];
    const int tile_m = wgid / g.tiles_n, tile_n = wgid - tile_m * g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t pitch = (size_t)g.K * 6;
    // 32-bit byte offsets from the tile's first row (256 rows x <= 24 KB); rows past M / N re-read the last valid row
    unsigned offA[LA], offW[2][LH];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        const int slot = tid + NT * i, r = slot / U, cp = slot % U;
        const int rr = FULL ? r : min(m0 + r, g.M - 1) - m0;
        offA[i] = (unsigned)bf3_row_offset(rr, g.K, g.epi.x_pair) + ((cp + 6 * ((r >> 3) & 1)) % U) * 16;        // relative to row m0 (even)
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < LH; i++) {
            const int slot = tid + NT * i, r = slot / U, cp = slot % U;
            const int rr = FULL ? h * HN + r : min(n0 + h * HN + r, g.N - 1) - n0;
            offW[h][i] = (unsigned)bf3_w_row_offset(rr, g.K) + ((cp + 6 * ((r >> 3) & 1)) % U) * 16;      // relative to row n0 (even)
        }
    const char* Abase = reinterpret_cast<const char*>(P.A) + bf3_row_offset(m0, g.K, g.epi.x_pair);
    const char* Wbase = reinterpret_cast<const char*>(P.Wt) + bf3_w_row_offset(n0, g.K);            // n0 is a multiple of 256
    auto issue_a = [&](int kt, int buf) {
        char* base = As + buf * A_BYTES + wave * 1024;
        const char* src = Abase + (size_t)kt * (g.epi.x_pair ? 384 : 192);
#pragma unroll
        for (int i = 0; i < LA; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + offA[i]), (lptr_t)(base + NT * 16 * i), 16, 0, 0);
    };
    auto issue_w = [&](int kt, int half) {
        char* base = Hs + half * H_BYTES + wave * 1024;
        const char* src = Wbase + (size_t)kt * BF3_W_KBLOCK_BYTES;
#pragma unroll
        for (int i = 0; i < LH; i++) __builtin_amdgcn_global_load_lds((gptr_t)(src + offW[half][i]), (lptr_t)(base + NT * 16 * i), 16, 0, 0);
    };
    f32x4 acc[8][2][2];                     // [32-column pair][row tile][column tile of the pair]
#pragma unroll
    for (int cpair = 0; cpair < 8; cpair++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[cpair][i][j][e] = 0.f;
    const int frow = lane & 15, kg = lane >> 4, rotk = 2 * ((frow >> 3) & 1);
    int fo[3];
#pragma unroll
    for (int p = 0; p < 3; p++) fo[p] = (frow * U + 3 * ((kg - rotk + 4) % 4) + p) * 16;
    const int nk = g.K / 32;
    issue_a(0, 0);
    issue_w(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        wait_vmcnt<0>();                                // A_kt and W-half-0 of stage kt (this wave's share) have landed
        __builtin_amdgcn_s_barrier();                   // ... everyone's; and every wave is done with stage kt-1
        issue_w(kt, 1);                                 // half 1 of THIS stage (its buffer held half 1 of stage kt-1)
        if (kt + 1 < nk) issue_a(kt + 1, (kt + 1) & 1);
        const char* sa = As + (kt & 1) * A_BYTES + wave * 32 * U * 16;
        bf16x8 af[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int p = 0; p < 3; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(sa + fo[p] + i * 16 * U * 16);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half == 1) {
                if (kt + 1 < nk) wait_vmcnt<LA>(); else wait_vmcnt<0>();       // half 1 landed; A of stage kt+1 may stay in flight
                __builtin_amdgcn_s_barrier();                                   // everyone is done with half 0 of this stage
                if (kt + 1 < nk) issue_w(kt + 1, 0);
            }
            const char* sw = Hs + half * H_BYTES;
#pragma unroll
            for (int c = 0; c < 2; c++) {               // 4 column tiles at a time (fragment registers)
                bf16x8 bf[4][3];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int p = 0; p < 3; p++) bf[j][p] = *reinterpret_cast<const bf16x8*>(sw + fo[p] + (c * 4 + j) * 16 * U * 16);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        f32x4& a = acc[half * 4 + c * 2 + (j >> 1)][i][j & 1];
                        if (NP == 6) {
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], a, 0, 0, 0);
                        }
                        if (NP >= 3) {
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], a, 0, 0, 0);
                        }
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], a, 0, 0, 0);
                    }
            }
        }
    }
    // eight explicit calls (a loop here is not unrolled at this body size, which would send the accumulators to scratch)
#define A3R_W2H_EPI(CP) gemm_epilogue16<2, 2, FULL>(g, P, acc[CP], m0, n0, wave * 32, (CP) * 32, lane)
    A3R_W2H_EPI(0); A3R_W2H_EPI(1); A3R_W2H_EPI(2); A3R_W2H_EPI(3); A3R_W2H_EPI(4); A3R_W2H_EPI(5); A3R_W2H_EPI(6); A3R_W2H_EPI(7);
#undef A3R_W2H_EPI
}

// fp32 [M, ldx] -> bf3 [M][K/8][3][8]: one thread per group of 8 consecutive k (32 B in, 48 contiguous bytes out)
// WPAIR: the row-pair weight layout of bf3.h (K % 32 == 0)
template <bool WPAIR>
__global__ __launch_bounds__(256) void split_bf3_kernel(const float* __restrict__ x, int ldx, char* __restrict__ y, long M, int K8) {
    const long total = M * K8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / K8;
        const int kg = (int)(i - row * K8);
        const f32x4* src = reinterpret_cast<const f32x4*>(x + row * ldx + kg * 8);
        if (WPAIR) bf3_store8(y + bf3_w_row_offset((int)row, K8 * 8) + (size_t)(kg >> 2) * BF3_W_KBLOCK_BYTES, (kg & 3) * 8, src[0], src[1]);
        else bf3_store8(y + row * ((size_t)K8 * 48), kg * 8, src[0], src[1]);
    }
}

struct Bf3Tile { int bm, bn, occ; double eff; };
// Tile shapes and their main-loop efficiencies measured on MI355X (tools/gemm_bf3_lab.hip, 18432-row ViT-L shapes, all on
// the 16x16x32 MFMA): 256x128 with 16 waves (4x4, one workgroup per CU) ~205, 128x64 with 4 waves (two workgroups per CU)
// ~180-210, 64x64 ~150 TFLOP/s fp32-equivalent.  A launch of n workgroups takes ceil(n / (256 occ)) rounds.
// Tile 3 = the 256x256 "W in two halves" kernel (nn.Linear only): 2/3 of tile 0's delivery bytes per flop; measured equal to
// tile 0 per flop on the ViT-L shapes (its 256 KB-per-workgroup epilogue eats the gain), so it wins only through quantisation.
static const Bf3Tile kTiles[4] = {{256, 128, 1, 1.0}, {128, 64, 2, 0.95}, {64, 64, 2, 0.75}, {256, 256, 1, 0.99}};

static int choose_bf3_tile(int M, int N, int groups, bool linear) {
    const int ntiles = linear ? 4 : 3;
    if (const char* f = getenv("A3R_BF3_TILE")) {      // developer override: 0 | 1 | 2 | 3
        const int t = atoi(f);
        if (t >= 0 && t < ntiles) return t;
    }
    int best_t = 0;
    double best = 1e300;
    for (int t = 0; t < ntiles; t++) {
        const long n = (long)((M + kTiles[t].bm - 1) / kTiles[t].bm) * ((N + kTiles[t].bn - 1) / kTiles[t].bn) * groups;
        const long slots = 256L * kTiles[t].occ;
        const double cost = (double)((n + slots - 1) / slots) * kTiles[t].bm * kTiles[t].bn * kTiles[t].occ / kTiles[t].eff;
        if (cost < best * 0.999) { best = cost; best_t = t; }
    }
    return best_t;
}

// process-wide arithmetic mode of the bf3 kernels (a3r_bf3_set_products): 6 (fp32-accurate), 3 or 1
static int g_bf3_products = 6;
int bf3_products() { return g_bf3_products; }

template <int AMODE, int BM, int BN, int BK, int WM, int WN, int NS, int MF, bool FULL, int NP>
static int launch_bf3_variant(const GemmArgs& g, hipStream_t st) {
    auto kern = gemm_bf3_kernel<AMODE, BM, BN, BK, WM, WN, NS, MF, FULL, NP>;
    constexpr int lds = NS * (BM + BN) * (3 * BK / 8) * 16;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }));
    hipLaunchKernelGGL(kern, dim3(g.tiles_per_group * g.groups), dim3(WM * WN * 64), lds, st, g);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

template <bool FULL, int NP>
static int launch_bf3_w2h(const GemmArgs& g, hipStream_t st) {
    auto kern = gemm_bf3_w2h_kernel<FULL, NP>;
    constexpr int lds = 2 * 256 * 192 + 2 * 128 * 192;
    static PerDeviceOnce attr_once;
    A3R_HIP(attr_once.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }));
    hipLaunchKernelGGL(kern, dim3(g.tiles_per_group * g.groups), dim3(512), lds, st, g);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

template <int AMODE>
static int launch_bf3(GemmArgs& g, hipStream_t st) {
    static const bool direct_epi = getenv("A3R_BF3_DIRECT_EPI") != nullptr;
    g.direct_epilogue = direct_epi ? 1 : 0;
    const int t = choose_bf3_tile(g.M, g.N, g.groups, AMODE == 0);
    const int bm = kTiles[t].bm, bn = kTiles[t].bn;
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    g.tiles_per_group = g.tiles_m * g.tiles_n;
    const bool full = g.M % bm == 0 && g.N % bn == 0;
    // algorithmic bytes: A once (linear: M K 6; conv: the input map B H W Cin 6), W once, results / residuals once
    const double mn = (double)g.M * g.N;
    const double a_bytes = AMODE == 0 ? 6.0 * g.M * g.K : 6.0 * (g.M / (g.cHo * g.cWo)) * g.cH * g.cW * g.cCin;
    const double c_bytes = mn * ((g.epi.out_bf3 ? 6.0 : 4.0) + (g.epi.aux_bf3 ? 6.0 : 0.0) +
                                 (g.epi.epi == A3R_EPI_RESID ? 4.0 : g.epi.epi == A3R_EPI_RESID2 ? 8.0 : 0.0));
    ProfScope prof(AMODE == 0 ? PK_LINEAR_BF3 : PK_CONV_BF3, 2.0 * g.M * g.N * g.K * g.groups, st,
                   g.groups * (a_bytes + 6.0 * g.N * g.K + c_bytes));
#define A3R_BF3_DISPATCH(NPV)                                                                                                   \
    do {                                                                                                                        \
        if (t == 0) return full ? launch_bf3_variant<AMODE, 256, 128, 32, 4, 4, 2, 16, true, NPV>(g, st)                        \
                                : launch_bf3_variant<AMODE, 256, 128, 32, 4, 4, 2, 16, false, NPV>(g, st);                      \
        if (t == 1) return full ? launch_bf3_variant<AMODE, 128, 64, 32, 2, 2, 2, 16, true, NPV>(g, st)                         \
                                : launch_bf3_variant<AMODE, 128, 64, 32, 2, 2, 2, 16, false, NPV>(g, st);                       \
        return full ? launch_bf3_variant<AMODE, 64, 64, 32, 2, 2, 3, 16, true, NPV>(g, st)                                      \
                    : launch_bf3_variant<AMODE, 64, 64, 32, 2, 2, 3, 16, false, NPV>(g, st);                                    \
    } while (0)
    if (AMODE == 0 && t == 3) {
        const int np = g_bf3_products;
        if (full) return np == 6 ? launch_bf3_w2h<true, 6>(g, st) : np == 3 ? launch_bf3_w2h<true, 3>(g, st) : launch_bf3_w2h<true, 1>(g, st);
        return np == 6 ? launch_bf3_w2h<false, 6>(g, st) : np == 3 ? launch_bf3_w2h<false, 3>(g, st) : launch_bf3_w2h<false, 1>(g, st);
    }
    if (g_bf3_products == 6) A3R_BF3_DISPATCH(6);
    if (g_bf3_products == 3) A3R_BF3_DISPATCH(3);
    A3R_BF3_DISPATCH(1);
#undef A3R_BF3_DISPATCH
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_bf3_set_products(int products) {
    const int prev = g_bf3_products;
    if (products == 6 || products == 3 || products == 1) g_bf3_products = products;
    return prev;
}

extern "C" size_t a3r_bf3_bytes(long rows, int K) { return rows > 0 && K > 0 ? (size_t)rows * K * 6 : 0; }

static int split_bf3_impl(const float* x, int ldx, void* y, long M, int K, bool wpair, void* stream) {
    const char* who = wpair ? "a3r_split_bf3_w" : "a3r_split_bf3";
    A3R_CHECK_ARG(x && y, "%s: null pointer", who);
    A3R_CHECK_ARG(M > 0 && K > 0 && K % (wpair ? 32 : 8) == 0, "%s: K (%d) must be a positive multiple of %d (M=%ld)", who, K, wpair ? 32 : 8, M);
    A3R_CHECK_ARG(!wpair || M < (1L << 31), "%s: too many rows", who);
    A3R_CHECK_ARG(ldx >= K && ldx % 4 == 0, "%s: bad leading dimension %d", who, ldx);
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    const long total = M * (K / 8);
    hipStream_t st = as_stream(stream);
    ProfScope prof(PK_SPLIT, 10.0 * M * K, st);
    const long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4));
    if (wpair) hipLaunchKernelGGL(split_bf3_kernel<true>, grid, dim3(256), 0, st, x, ldx, static_cast<char*>(y), M, K / 8);
    else hipLaunchKernelGGL(split_bf3_kernel<false>, grid, dim3(256), 0, st, x, ldx, static_cast<char*>(y), M, K / 8);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_split_bf3(const float* x, int ldx, void* y, long M, int K, void* stream) {
    return split_bf3_impl(x, ldx, y, M, K, false, stream);
}

// weights [N, K] -> the row-pair weight layout (bf3.h); y holds a3r_bf3_w_bytes(N, K) bytes (an odd N leaves the last pair half empty)
extern "C" size_t a3r_bf3_w_bytes(long rows, int K) { return rows > 0 && K > 0 ? (size_t)((rows + 1) / 2 * 2) * K * 6 : 0; }
extern "C" int a3r_split_bf3_w(const float* w, int ldw, void* y, long N, int K, void* stream) {
    return split_bf3_impl(w, ldw, y, N, K, true, stream);
}

extern "C" int a3r_linear_bf3_grouped(const a3r_group_ptrs_bf3* groups, int n_groups, int ldc, int M, int N, int K,
                                      const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(groups && n_groups >= 1 && n_groups <= 4, "a3r_linear_bf3_grouped: 1..4 groups required");
    A3R_CHECK_ARG(M > 0 && N > 0 && K > 0, "a3r_linear_bf3: M, N, K must be positive (got %d, %d, %d)", M, N, K);
    A3R_CHECK_ARG(K % 32 == 0, "a3r_linear_bf3: K (%d) must be a multiple of 32", K);
    A3R_CHECK_ARG(ldc >= 1, "a3r_linear_bf3: bad leading dimension ldc=%d", ldc);
    if (int rc = check_epilogue(epi, M, N, "a3r_linear_bf3", true)) return rc;
    GemmArgs g = {};
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(!g.epi.relu_a, "a3r_linear_bf3: relu_a is only available on a3r_conv3x3");
    for (int i = 0; i < n_groups; i++) {
        g.grp[i] = {static_cast<const float*>(groups[i].x3), static_cast<const float*>(groups[i].w3), groups[i].y, groups[i].bias,
                    groups[i].resid, groups[i].resid2};
        if (int rc = check_group(g.grp[i], g.epi.epi, "a3r_linear_bf3")) return rc;
    }
    g.groups = n_groups;
    g.lda = K; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    if (g.epi.epi != A3R_EPI_PIXSHUF) A3R_CHECK_ARG(ldc >= N, "a3r_linear_bf3: ldc (%d) < N (%d)", ldc, N);
    if (g.epi.out_bf3) A3R_CHECK_ARG(ldc == N, "a3r_linear_bf3: out_bf3 needs ldc == N");
    return launch_bf3<0>(g, as_stream(stream));
}

extern "C" int a3r_linear_bf3(const void* x3, const void* w3, float* y, int ldc, int M, int N, int K, const a3r_epilogue* epi,
                              void* stream) {
    a3r_group_ptrs_bf3 p = {x3, w3, y, epi ? epi->bias : nullptr, epi ? epi->resid : nullptr, epi ? epi->resid2 : nullptr};
    return a3r_linear_bf3_grouped(&p, 1, ldc, M, N, K, epi, stream);
}

extern "C" int a3r_conv3x3_bf3(const void* x3, const void* wp3, float* y, int B, int H, int W, int Cin, int Cout, int stride,
                               const a3r_epilogue* epi, void* stream) {
    A3R_CHECK_ARG(x3 && wp3 && y, "a3r_conv3x3_bf3: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0, "a3r_conv3x3_bf3: bad shape");
    A3R_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "a3r_conv3x3_bf3: Cin (%d) must be a multiple of 32", Cin);
    A3R_CHECK_ARG(stride == 1 || stride == 2, "a3r_conv3x3_bf3: stride must be 1 or 2");
    GemmArgs g = {};
    g.cH = H; g.cW = W; g.cCin = Cin; g.cStride = stride;
    g.cHo = (H + 2 - 3) / stride + 1;
    g.cWo = (W + 2 - 3) / stride + 1;
    g.lda = 0; g.ldc = Cout;
    g.M = B * g.cHo * g.cWo; g.N = Cout; g.K = 9 * Cin;
    if (int rc = check_epilogue(epi, g.M, g.N, "a3r_conv3x3_bf3", true)) return rc;
    if (epi) g.epi = *epi;
    A3R_CHECK_ARG(g.epi.epi != A3R_EPI_PIXSHUF && g.epi.epi != A3R_EPI_ROPE, "a3r_conv3x3_bf3: unsupported epilogue");
    A3R_CHECK_ARG(!g.epi.relu_a, "a3r_conv3x3_bf3: relu_a is not available (the producer writes the pre-activated bf3 input: aux_relu)");
    A3R_CHECK_ARG(!g.epi.x_pair && !g.epi.out_pair, "a3r_conv3x3_bf3: the row-pair layout is for a3r_linear_bf3 operands only");
    g.groups = 1;
    g.grp[0] = {static_cast<const float*>(x3), static_cast<const float*>(wp3), y, g.epi.bias, g.epi.resid, g.epi.resid2};
    if (int rc = check_group(g.grp[0], g.epi.epi, "a3r_conv3x3_bf3")) return rc;
    return launch_bf3<1>(g, as_stream(stream));
}
