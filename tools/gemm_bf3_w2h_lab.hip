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

// 256 x 256 output tile, v_mfma_f32_16x16x32_bf16, BK = 32.  Eight waves, each a 32-row strip over ALL 256 columns
// (acc 2 x 16 tiles = 128 VGPRs).  LDS: A stages double-buffered (2 x 48 KB); W ONE stage in two column halves (2 x 24 KB):
// a stage multiplies the strip by W half 0, then by half 1; half 0 is refilled for the next stage at the mid-stage barrier,
// half 1 and the next A stage at the stage barrier.  Delivery bytes per flop are 2/3 of the 256 x 128 tile's.
__global__ __launch_bounds__(512) void gemm_w2h(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, float* __restrict__ C,
                                                int M, int N, int K) {
    constexpr int BM = 256, BN = 256, HN = 128, NT = 512, U = 12;
    constexpr int A_BYTES = BM * U * 16, H_BYTES = HN * U * 16;          // 49152, 24576
    constexpr int LA = BM * U / NT, LH = HN * U / NT;                      // 6, 3 DMAs per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [2][256][12 units]
    char* Hs = smem + 2 * A_BYTES;          // [2 halves][128][12 units]
    const int tiles_n = N / BN, tiles_m = M / BM, nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile_m = wgid / tiles_n, tile_n = wgid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t pitch = (size_t)K * 6;
    // 32-bit byte offsets from the operand bases (row pitch <= 24 KB, 256 rows: < 8 MB)
    unsigned offA[LA], offW[LH];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        const int slot = tid + NT * i, r = slot / U, cp = slot % U;
        offA[i] = (unsigned)(r * pitch) + ((cp + 6 * ((r >> 3) & 1)) % U) * 16;
    }
#pragma unroll
    for (int i = 0; i < LH; i++) {
        const int slot = tid + NT * i, r = slot / U, cp = slot % U;
        offW[i] = (unsigned)(r * pitch) + ((cp + 6 * ((r >> 3) & 1)) % U) * 16;
    }
    const char* Abase = reinterpret_cast<const char*>(Ap) + (size_t)m0 * pitch;
    const char* Wbase = reinterpret_cast<const char*>(Wp) + (size_t)n0 * pitch;
    auto issue_a = [&](int kt, int buf) {
        char* base = As + buf * A_BYTES + wave * 1024;
        const char* g = Abase + (size_t)kt * 192;
#pragma unroll
        for (int i = 0; i < LA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + offA[i]),
                                             (__attribute__((address_space(3))) void*)(base + NT * 16 * i), 16, 0, 0);
    };
    auto issue_w = [&](int kt, int half) {
        char* base = Hs + half * H_BYTES + wave * 1024;
        const char* g = Wbase + (size_t)half * HN * pitch + (size_t)kt * 192;
#pragma unroll
        for (int i = 0; i < LH; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + offW[i]),
                                             (__attribute__((address_space(3))) void*)(base + NT * 16 * i), 16, 0, 0);
    };
    f32x4 acc[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 16; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
    const int frow = lane & 15, kg = lane >> 4, rotk = 2 * ((frow >> 3) & 1);
    int fo[3];
#pragma unroll
    for (int p = 0; p < 3; p++) fo[p] = (frow * U + 3 * ((kg - rotk + 4) % 4) + p) * 16;
    const int nk = K / 32;
    issue_a(0, 0);
    issue_w(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        wait_vm<0>();                                   // A_kt and W0_kt (this wave's share) have landed
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
                if (kt + 1 < nk) wait_vm<LA>(); else wait_vm<0>();     // W1_kt landed; A_{kt+1} may stay in flight
                __builtin_amdgcn_s_barrier();                           // everyone is done with half 0 of this stage
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
                        f32x4& a = acc[i][half * 8 + c * 4 + j];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], a, 0, 0, 0);
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int col = n0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = m0 + wave * 32 + i * 16 + (lane >> 4) * 4 + e;
                C[(size_t)row * N + col] = acc[i][j][e];
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

double run(const char* name, const uint16_t* Ap, const uint16_t* Wp, float* C, int M, int N, int K, int iters) {
    auto kern = gemm_w2h;
    const int lds = 2 * 49152 + 2 * 24576;
    if (M % 256 || N % 256 || K % 32) { printf("%-34s skipped (shape)\n", name); return 0; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = (M / 256) * (N / 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, Ap, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s M=%5d N=%5d K=%5d blocks=%5d lds=%6d  %8.1f us  %7.2f TF(fp32-equiv)\n", name, M, N, K, blocks, lds, us, tf);
    return us;
}

int main(int argc, char** argv) {
    const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
    const int shapes[][3] = {{4096, 4096, 4096}, {18432, 4096, 1024}, {18432, 1024, 1024}, {64512, 4096, 1024}, {64512, 1024, 1024},
                             {64512, 768, 768}, {64512, 1024, 4096}, {64512, 3072, 1024}};
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
        if (run("w2h 256x256 8w m16", Ap, Wp, C, M, N, K, 20) > 0) {
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
            printf("   max |err| / sum|a||b| = %.3e\n", maxrel);
        }
        CK(hipFree(Ap)); CK(hipFree(Wp)); CK(hipFree(C));
    }
    return 0;
}
