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

__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
        case 0: wait_vm<0>(); break;  case 1: wait_vm<1>(); break;  case 2: wait_vm<2>(); break;  case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;  case 5: wait_vm<5>(); break;  case 6: wait_vm<6>(); break;  case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break;  case 9: wait_vm<9>(); break;  case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; case 15: wait_vm<15>(); break;
        case 16: wait_vm<16>(); break; case 18: wait_vm<18>(); break; case 20: wait_vm<20>(); break; case 24: wait_vm<24>(); break;
        default: wait_vm<0>(); break;
    }
}

// Ping-pong ("two K-parity groups") bf3 GEMM: the workgroup's waves form two groups that both cover the whole BM x BN tile;
// group g owns the stages of parity g.  In time slot t group t&1 READS the fragments of stage t into registers while the other
// group runs the MFMAs of stage t-1 from its registers, so LDS reads of one group overlap the matrix work of the other.
// One barrier per slot; the buffer of stage t-1 is refilled (stage t-1+NS) right after the barrier of slot t.
// MF = 16: v_mfma_f32_16x16x32_bf16, BK = 32.  MF = 32: v_mfma_f32_32x32x16_bf16, BK = 16.
template <int BM, int BN, int MF, int WM, int WN, int NS>
__global__ __launch_bounds__(2 * WM * WN * 64) void gemm_pp(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, float* __restrict__ C,
                                                             int M, int N, int K) {
    constexpr bool M16 = MF == 16;
    constexpr int BK = M16 ? 32 : 16;
    constexpr int GW = WM * WN, NW = 2 * GW, NT = NW * 64, KG = BK / 8, U = 3 * KG;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int SA = BM * U, SB = BN * U;
    constexpr int LA = (SA + NT - 1) / NT, LB = (SB + NT - 1) / NT;
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
    const int grp = wave / GW, wl = wave % GW, wm = wl / WN, wn = wl % WN;
    const size_t pitch = (size_t)K * 6;
    auto src_unit = [&](int r, int cp) { return M16 ? (cp + 6 * ((r >> 3) & 1)) % U : (cp + (r / (16 / KG)) % KG) % U; };
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
    const bool lastA = (LA - 1) * NT + wave * 64 < SA, lastB = (LB - 1) * NT + wave * 64 < SB;
    const int lps = LA + LB - ((SA % NT != 0 && !lastA) ? 1 : 0) - ((SB % NT != 0 && !lastB) ? 1 : 0);
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * STAGE + wave * 1024;
        const size_t koff = (size_t)kt * (KG * 48);
#pragma unroll
        for (int i = 0; i < LA; i++)
            if (i + 1 < LA || SA % NT == 0 || lastA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(base + NT * 16 * i), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; i++)
            if (i + 1 < LB || SB % NT == 0 || lastB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(base + SA * 16 + NT * 16 * i), 16, 0, 0);
    };
    constexpr int TM = WTM / MF, TN = WTN / MF, AE = M16 ? 4 : 16;
    typedef float accv __attribute__((ext_vector_type(AE)));
    accv acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < AE; e++) acc[i][j][e] = 0.f;
    int offA[3], offB[3];
    if (M16) {
        const int frow = lane & 15, kg = lane >> 4, rotk = 2 * ((frow >> 3) & 1);
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int c = 3 * ((kg - rotk + 4) % 4) + p;
            offA[p] = ((wm * WTM + frow) * U + c) * 16;
            offB[p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
        }
    } else {
        const int frow = lane & 31, fh = lane >> 5, rot = (frow / (16 / KG)) % KG;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int c = (fh * 3 + p - rot + U) % U;
            offA[p] = ((wm * WTM + frow) * U + c) * 16;
            offB[p] = SA * 16 + ((wn * WTN + frow) * U + c) * 16;
        }
    }
    bf16x8 af[TM][3], bf[TN][3];
    const int nk = K / BK;
    int issued = 0;
    for (; issued < NS && issued < nk; issued++) issue(issued, issued % NS);
    for (int t = 0; t <= nk; t++) {
        if (t < nk) wait_vm_dyn((issued - 1 - t) * lps);        // this wave's DMAs of stage t have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments read in the previous slot are in registers
        __builtin_amdgcn_s_barrier();
        if (t >= 1 && issued < nk) { issue(issued, issued % NS); issued++; }     // refills the buffer of stage t-1
        if (grp == (t & 1)) {
            if (t < nk) {
                const char* sb = smem + (t % NS) * STAGE;
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int p = 0; p < 3; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(sb + offA[p] + i * MF * U * 16);
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int p = 0; p < 3; p++) bf[j][p] = *reinterpret_cast<const bf16x8*>(sb + offB[p] + j * MF * U * 16);
            }
        } else if (t >= 1) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if constexpr (M16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    }
                }
        }
    }
    // ---- fold the two K-parity halves: group 1 parks its accumulators in LDS, group 0 adds them and stores
    __syncthreads();
    float* park = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < AE; e++) park[((i * TN + j) * AE + e) * (GW * 64) + wl * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < AE; e++) {
                    const float v = acc[i][j][e] + park[((i * TN + j) * AE + e) * (GW * 64) + wl * 64 + lane];
                    int row, col;
                    if (M16) { row = m0 + wm * WTM + i * 16 + (lane >> 4) * 4 + e; col = n0 + wn * WTN + j * 16 + (lane & 15); }
                    else { row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); col = n0 + wn * WTN + j * 32 + (lane & 31); }
                    C[(size_t)row * N + col] = v;
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

template <int BM, int BN, int MF, int WM, int WN, int NS>
double run(const char* name, const uint16_t* Ap, const uint16_t* Wp, float* C, int M, int N, int K, int iters) {
    auto kern = gemm_pp<BM, BN, MF, WM, WN, NS>;
    constexpr int BK = MF == 16 ? 32 : 16;
    const int lds = NS * (BM + BN) * (3 * BK / 8) * 16;
    if (M % BM || N % BN || K % BK || lds > 160 * 1024) { printf("%-34s skipped (shape/lds %d)\n", name, lds); return 0; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = (M / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(2 * WM * WN * 64), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(2 * WM * WN * 64), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s M=%5d N=%5d K=%5d blocks=%5d lds=%6d  %8.1f us  %7.2f TF(fp32-equiv)\n", name, M, N, K, blocks, lds, us, tf);
    return us;
}

int main(int argc, char** argv) {
    const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
    const unsigned mask = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0xffffffffu;
    const int shapes[][3] = {{4096, 4096, 4096}, {18432, 4096, 1024}, {18432, 3072, 1024}, {18432, 1024, 1024}, {18432, 1024, 4096},
                             {18432, 768, 768}, {18432, 768, 3072}};
    int shape_idx = -1;
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
        CK(hipMalloc(&Ap, hAp.size() * 2)); CK(hipMalloc(&Wp, hWp.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
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
#define RUN(name, ...) do { if (mask & (1u << bit)) { if (run<__VA_ARGS__>(name, Ap, Wp, C, M, N, K, it) > 0) check(name); } bit++; } while (0)
        RUN("pp 128x128 m16 2x(2x2) NS3", 128, 128, 16, 2, 2, 3);
        RUN("pp 256x128 m32 2x(4x2) NS3", 256, 128, 32, 4, 2, 3);
        RUN("pp 256x128 m32 2x(4x2) NS4", 256, 128, 32, 4, 2, 4);
        RUN("pp 128x128 m32 2x(2x2) NS4", 128, 128, 32, 2, 2, 4);
        RUN("pp 128x128 m32 2x(2x2) NS6", 128, 128, 32, 2, 2, 6);
        RUN("pp 256x128 m16 2x(4x2) NS2", 256, 128, 16, 4, 2, 2);
        CK(hipFree(Ap)); CK(hipFree(Wp)); CK(hipFree(C));
    }
    return 0;
}
