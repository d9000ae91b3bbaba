// Developer lab (not part of the product): fp32-accurate GEMM on the bf16 matrix cores, both operands PRE-SPLIT into
// three bf16 planes ("bf3": [rows][K/8][3][8] bf16, x = x0+x1+x2 exactly), staged global->LDS by LDS-DMA (glds).
//   C = sum over (p,q), p+q<=2 of A_p * W_q^T   (six exact-product MFMA passes, fp32 accumulate)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bf3_lab.hip -o build/gemm_bf3_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image of one operand stage: rows of U = 3*BK/8 16-byte units, unpadded, with the units of each row rotated on the
// source side so that the MFMA operand reads (ds_read_b128) are conflict-free.
// VAR bits: 1 = s_setprio(1) around the MFMA cluster; 2 = v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (BK must be 32);
// 4 = no DMA after the prologue; 8 = only the a0b0 product; 16 = no MFMA at all; 32 = TIMING ONLY: sources addressed as if both
// operands were stored pre-tiled (a stage of a tile = one contiguous block), results are meaningless
template <int BM, int BN, int BK, int WM, int WN, int NS, int VAR>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf3(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, float* __restrict__ C,
                                                          int M, int N, int K) {
    constexpr bool M16 = (VAR & 2) != 0;
    constexpr int NT = WM * WN * 64, KG = BK / 8, U = 3 * KG, G = KG, PER = 16 / G;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int SA = BM * U, SB = BN * U;
    constexpr int LA = (SA + NT - 1) / NT, LB = (SB + NT - 1) / NT;       // the last chunk may cover only the first waves
    static_assert(SA % 64 == 0 && SB % 64 == 0, "whole waves");
    constexpr int STAGE = (SA + SB) * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = N / BN, tiles_m = M / BM, nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile_m = wgid / tiles_n, tile_n = wgid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const size_t pitch = (size_t)K * 6 + ((VAR & 512) ? 128 : 0);   // bytes per bf3 row (VAR & 512: one extra cache line, so rows start on rotating L2 channels)
    // source-side rotation: 32x32x16 operands rotate single units by (r / PER) % G; 16x16x32 operands (BK = 32) rotate whole
    // k-groups (3 units) by 2 * ((r >> 3) & 1)
    // VAR & 4096: the rotation is baked into the storage (rows with bit 3 set keep their k-groups rotated by two), the DMA is lane-linear
    auto src_unit = [&](int r, int cp) { return (VAR & 4096) ? cp : M16 ? (cp + 6 * ((r >> 3) & 1)) % U : (cp + (r / PER) % G) % U; };
    const char* srcA[LA];
    const char* srcB[LB];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        const int slot = tid + NT * i, r = (slot / U) % BM, cp = slot % U;
        srcA[i] = reinterpret_cast<const char*>(Ap) + (size_t)(m0 + r) * pitch + src_unit(r, cp) * 16;
    }
#pragma unroll
    for (int i = 0; i < LB; i++) {
        const int slot = tid + NT * i, r = (slot / U) % BN, cp = slot % U;
        srcB[i] = reinterpret_cast<const char*>(Wp) + (size_t)(n0 + r) * pitch + src_unit(r, cp) * 16;
    }
    // row-block interleaved storage [R/RB][K/32][RB rows][12 units]: a block's k-slice is RB*192 contiguous, line-aligned bytes.
    // VAR & 64: RB = 2 (both operands); VAR & 128: RB = 16 (both); VAR & 256: RB = 16 for W only (A stays row-major)
    constexpr int RBA = (VAR & 64) ? 2 : (VAR & 128) ? 16 : (VAR & 1024) ? 64 : (VAR & 2048) ? 256 : 1;
    constexpr int RBB = (VAR & 64) ? 2 : (VAR & (128 | 256)) ? 16 : (VAR & 1024) ? 64 : (VAR & 2048) ? 128 : 1;
    if (RBA > 1) {
#pragma unroll
        for (int i = 0; i < LA; i++) {
            const int slot = tid + NT * i, r = (slot / U) % BM, cp = slot % U, gm = m0 + r;
            srcA[i] = reinterpret_cast<const char*>(Ap) + (size_t)(gm / RBA) * (RBA * pitch) + (gm % RBA) * 192 + src_unit(r, cp) * 16;
        }
    }
    if (RBB > 1) {
#pragma unroll
        for (int i = 0; i < LB; i++) {
            const int slot = tid + NT * i, r = (slot / U) % BN, cp = slot % U, gn = n0 + r;
            srcB[i] = reinterpret_cast<const char*>(Wp) + (size_t)(gn / RBB) * (RBB * pitch) + (gn % RBB) * 192 + src_unit(r, cp) * 16;
        }
    }
    if (VAR & 32) {
        const int nkk = K / BK;
#pragma unroll
        for (int i = 0; i < LA; i++) srcA[i] = reinterpret_cast<const char*>(Ap) + ((size_t)tile_m * nkk * SA + (size_t)((tid + NT * i) % SA)) * 16;
#pragma unroll
        for (int i = 0; i < LB; i++) srcB[i] = reinterpret_cast<const char*>(Wp) + ((size_t)tile_n * nkk * SB + (size_t)((tid + NT * i) % SB)) * 16;
    }
    const bool lastA = (LA - 1) * NT + wave * 64 < SA, lastB = (LB - 1) * NT + wave * 64 < SB;   // wave-uniform
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * STAGE + wave * 1024;
        const size_t koffA = (size_t)kt * (KG * 48) * RBA, koffB = (size_t)kt * (KG * 48) * RBB;
        if (VAR & 32) {
#pragma unroll
            for (int i = 0; i < LA; i++)
                if (i + 1 < LA || SA % NT == 0 || lastA)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (size_t)kt * SA * 16),
                                                     (__attribute__((address_space(3))) void*)(base + NT * 16 * i), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < LB; i++)
                if (i + 1 < LB || SB % NT == 0 || lastB)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (size_t)kt * SB * 16),
                                                     (__attribute__((address_space(3))) void*)(base + SA * 16 + NT * 16 * i), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < LA; i++)
            if (i + 1 < LA || SA % NT == 0 || lastA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + koffA),
                                                 (__attribute__((address_space(3))) void*)(base + NT * 16 * i), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; i++)
            if (i + 1 < LB || SB % NT == 0 || lastB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + koffB),
                                                 (__attribute__((address_space(3))) void*)(base + SA * 16 + NT * 16 * i), 16, 0, 0);
    };
    // per-wave loads per stage (wave-uniform)
    const int lps = LA + LB - ((SA % NT != 0 && !lastA) ? 1 : 0) - ((SB % NT != 0 && !lastB) ? 1 : 0);
    auto wait_stages = [&](int stages_in_flight) {     // wait until at most `stages_in_flight` stages of this wave's DMAs are outstanding
        const int n = stages_in_flight * lps;
        // vmcnt takes an immediate: dispatch over the few possible values
        switch (n) {
            case 0: wait_vm<0>(); break;
            case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break; case 4: wait_vm<4>(); break;
            case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break; case 8: wait_vm<8>(); break;
            case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break; case 12: wait_vm<12>(); break;
            case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; case 15: wait_vm<15>(); break; case 16: wait_vm<16>(); break;
            case 17: wait_vm<17>(); break; case 18: wait_vm<18>(); break; case 20: wait_vm<20>(); break; case 24: wait_vm<24>(); break;
            default: wait_vm<0>(); break;
        }
    };
    const int nk = K / BK;
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
            const int rem = nk - 1 - kt;
            wait_stages(rem < NS - 2 ? rem : NS - 2);
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
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
                if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    }
                if (VAR & 1) __builtin_amdgcn_s_setprio(0);
            }
        }
        const int half = lane >> 5, lcol = lane & 31;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * WTN + j * 32 + lcol;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    C[(size_t)row * N + col] = acc[i][j][e];
                }
        }
    } else {
        static_assert(!M16 || BK == 32, "16x16x32 needs BK = 32");
        constexpr int TM = WTM / 16, TN = WTN / 16;
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
        // lane (row = lane & 15, k-group = lane >> 4) reads, per plane p, unit 3 * ((kg - rot) mod 4) + p
        const int frow = lane & 15, kg = lane >> 4, rotk = 2 * ((frow >> 3) & 1);
        int offA[3], offB[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int c = 3 * ((kg - rotk + 4) % 4) + p;
            offA[p] = ((wm * WTM + frow) * U + c) * 16;
            offB[p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
        }
        for (int kt = 0; kt < nk; kt++) {
            const int rem = nk - 1 - kt;
            wait_stages(rem < NS - 2 ? rem : NS - 2);
            __builtin_amdgcn_s_barrier();
            if (!(VAR & 4) && kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
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
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (!(VAR & 24)) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    }
                    if (!(VAR & 16)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    else {      // no MFMA at all: keep the fragment reads alive with one VALU op per fragment
#pragma unroll
                        for (int p = 0; p < 3; p++) acc[i][j][0] += (float)af[i][p][0] + (float)bf[j][p][0];
                    }
                }
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
        }
        // C/D of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * WTN + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int row = m0 + wm * WTM + i * 16 + (lane >> 4) * 4 + e;
                    C[(size_t)row * N + col] = acc[i][j][e];
                }
        }
    }
}

static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
    return (uint16_t)u;
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split_rows(const std::vector<float>& X, std::vector<uint16_t>& P, int R, int K) {
    P.resize((size_t)R * K * 3);
    for (int n = 0; n < R; n++)
        for (int k = 0; k < K; k++) {
            float x = X[(size_t)n * K + k];
            uint16_t p0 = bf16_rne(x); float r1 = x - bf16_f(p0);
            uint16_t p1 = bf16_rne(r1); float r2 = r1 - bf16_f(p1);
            uint16_t p2 = bf16_rne(r2);
            size_t base = ((size_t)n * (K / 8) + k / 8) * 24 + (k % 8);
            P[base] = p0; P[base + 8] = p1; P[base + 16] = p2;
        }
}

static void split_rows_blk(const std::vector<float>& X, std::vector<uint16_t>& P, int R, int K, int RB, int pad = 0, bool baked = false) {
    const size_t pitch = (size_t)6 * K + pad;
    P.assign((size_t)((R + RB - 1) / RB) * RB * pitch / 2, 0);
    for (int n = 0; n < R; n++)
        for (int k = 0; k < K; k++) {
            float x = X[(size_t)n * K + k];
            uint16_t p0 = bf16_rne(x); float r1 = x - bf16_f(p0);
            uint16_t p1 = bf16_rne(r1); float r2 = r1 - bf16_f(p1);
            uint16_t p2 = bf16_rne(r2);
            // bytes: (n/RB) * RB*6K + (k>>5) * RB*192 + (n%RB) * 192 + ((k>>3)&3) * 48 + plane * 16 + (k&7) * 2
            size_t base = ((size_t)(n / RB) * RB * pitch + (size_t)(k >> 5) * RB * 192 + (n % RB) * 192 + ((((k >> 3) & 3) + (baked && ((n >> 3) & 1) ? 2 : 0)) & 3) * 48) / 2 + (k % 8);
            P[base] = p0; P[base + 8] = p1; P[base + 16] = p2;
        }
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int VAR = 0>
double run(const char* name, const uint16_t* Ap, const uint16_t* Wp, float* C, int M, int N, int K, int iters) {
    auto kern = gemm_bf3<BM, BN, BK, WM, WN, NS, VAR>;
    const int lds = NS * (BM + BN) * (3 * BK / 8) * 16;
    if (M % BM || N % BN || K % BK) { printf("%-34s skipped (shape)\n", name); return 0; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = (M / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s M=%5d N=%5d K=%5d blocks=%5d lds=%6d  %8.1f us  %7.2f TF(fp32-equiv)\n", name, M, N, K, blocks, lds, us, tf);
    return us;
}

int main(int argc, char** argv) {
    const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
    const unsigned mask = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0xffffffffu;
    int shape_idx = -1;
    const int shapes[][3] = {{4096, 4096, 4096}, {18432, 4096, 1024}, {18432, 3072, 1024}, {18432, 1024, 1024}, {18432, 1024, 4096},
                             {18432, 768, 768}, {18432, 768, 3072}, {64512, 4096, 1024}, {64512, 1024, 1024}, {64512, 768, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        shape_idx++;
        if (only_shape >= 0 && shape_idx != only_shape) continue;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() * 0.05f;
        std::vector<uint16_t> hAp, hWp;
        split_rows(hA, hAp, M, K); split_rows(hW, hWp, N, K);
        uint16_t *Ap, *Wp; float* C;
        CK(hipMalloc(&Ap, hAp.size() * 2 + (size_t)(M + 16) * 128)); CK(hipMalloc(&Wp, hWp.size() * 2 + (size_t)(N + 16) * 128)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMemcpy(Ap, hAp.data(), hAp.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
        const int it = 20;
        auto check = [&](const char* what) {
            std::vector<float> hC((size_t)8 * N);
            CK(hipMemcpy(hC.data(), C + (size_t)(M - 8) * N, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxrel = 0;
            for (int rr = 0; rr < 8; rr++)
                for (int n = 0; n < N; n += 7) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; k++) { double p = (double)hA[(size_t)(M - 8 + rr) * K + k] * (double)hW[(size_t)n * K + k]; ref += p; mag += fabs(p); }
                    double e = fabs((double)hC[(size_t)rr * N + n] - ref) / mag;
                    if (e > maxrel) maxrel = e;
                }
            printf("   %s: max |err| / sum|a||b| = %.3e\n", what, maxrel);
            CK(hipMemset(C, 0, (size_t)M * N * 4));
        };
        int bit = 0;
#define RUN(name, ...) do { if (mask & (1u << bit)) { run<__VA_ARGS__>(name, Ap, Wp, C, M, N, K, it); check(name); } bit++; } while (0)
        RUN("256x128 16w m16 NS2", 256, 128, 32, 4, 4, 2, 2);
        RUN("256x128 16w m16 NS2 TILED-SRC", 256, 128, 32, 4, 4, 2, 2 | 32);
        auto upload = [&](int rba, int rbb, int pad = 0, bool baked = false) {
            std::vector<uint16_t> hA2, hW2;
            split_rows_blk(hA, hA2, M, K, rba, pad, baked); split_rows_blk(hW, hW2, N, K, rbb, pad, baked);
            CK(hipMemcpy(Ap, hA2.data(), hA2.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(Wp, hW2.data(), hW2.size() * 2, hipMemcpyHostToDevice));
        };
        upload(2, 2);
        RUN("256x128 16w m16 NS2 RB2", 256, 128, 32, 4, 4, 2, 2 | 64);
        RUN("256x128 16w m16 NS2 RB2 setprio", 256, 128, 32, 4, 4, 2, 2 | 64 | 1);
        RUN("256x128 8w(4x2) m16 NS2 RB2", 256, 128, 32, 4, 2, 2, 2 | 64);
        RUN("256x128 8w(2x4) m16 NS2 RB2", 256, 128, 32, 2, 4, 2, 2 | 64);
        upload(16, 16);
        RUN("256x128 16w m16 NS2 RB16", 256, 128, 32, 4, 4, 2, 2 | 128);
        RUN("128x64 4w m16 NS2 RB16", 128, 64, 32, 2, 2, 2, 2 | 128);
        RUN("64x64 4w m16 NS3 RB16", 64, 64, 32, 2, 2, 3, 2 | 128);
        upload(1, 16);
        RUN("256x128 16w m16 NS2 RB16 W only", 256, 128, 32, 4, 4, 2, 2 | 256);
        upload(1, 1, 0, true);
        RUN("256x128 16w m16 NS2 RB1 BAKED", 256, 128, 32, 4, 4, 2, 2 | 4096);
        upload(2, 2, 0, true);
        RUN("256x128 16w m16 NS2 RB2 BAKED", 256, 128, 32, 4, 4, 2, 2 | 64 | 4096);
        RUN("128x64 4w m16 NS2 RB2 BAKED", 128, 64, 32, 2, 2, 2, 2 | 64 | 4096);
        RUN("64x64 4w m16 NS3 RB2 BAKED", 64, 64, 32, 2, 2, 3, 2 | 64 | 4096);
        upload(16, 16, 0, true);
        RUN("256x128 16w m16 NS2 RB16 BAKED", 256, 128, 32, 4, 4, 2, 2 | 128 | 4096);
        upload(256, 128, 0, true);
        RUN("256x128 16w m16 NS2 RB256/128 BAKED", 256, 128, 32, 4, 4, 2, 2 | 2048 | 4096);
        upload(64, 64);
        RUN("256x128 16w m16 NS2 RB64", 256, 128, 32, 4, 4, 2, 2 | 1024);
        upload(256, 128);
        RUN("256x128 16w m16 NS2 RB256/128", 256, 128, 32, 4, 4, 2, 2 | 2048);
        upload(1, 1, 128);
        RUN("256x128 16w m16 NS2 PAD128", 256, 128, 32, 4, 4, 2, 2 | 512);
        RUN("128x64 4w m16 NS2 PAD128", 128, 64, 32, 2, 2, 2, 2 | 512);
        RUN("64x64 4w m16 NS3 PAD128", 64, 64, 32, 2, 2, 3, 2 | 512);
        upload(2, 2, 128);
        RUN("256x128 16w m16 NS2 RB2 PAD128", 256, 128, 32, 4, 4, 2, 2 | 64 | 512);
        upload(1, 1);
        RUN("128x64 4w m16 NS2 (row-major)", 128, 64, 32, 2, 2, 2, 2);
        RUN("64x64 4w m16 NS3 (row-major)", 64, 64, 32, 2, 2, 3, 2);
        RUN("256x128 16w m16 NS2 noMFMA", 256, 128, 32, 4, 4, 2, 2 | 16);
        RUN("256x128 16w m16 NS2 noMFMA TILED-SRC", 256, 128, 32, 4, 4, 2, 2 | 16 | 32);
        RUN("256x256x16 8w(4x2) m32 NS3", 256, 256, 16, 4, 2, 3, 0);
        RUN("256x256x16 8w(2x4) m32 NS3", 256, 256, 16, 2, 4, 3, 0);
        RUN("256x256x16 16w(4x4) m32 NS3", 256, 256, 16, 4, 4, 3, 0);
        RUN("128x64 4w m16 NS2", 128, 64, 32, 2, 2, 2, 2);
        CK(hipFree(Ap)); CK(hipFree(Wp)); CK(hipFree(C));
    }
    return 0;
}
