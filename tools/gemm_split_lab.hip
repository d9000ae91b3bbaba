// Developer lab (not part of the product): fp32 GEMM on the bf16 matrix cores by exact 3-way operand splitting.
//   x = x0 + x1 + x2 exactly (three round-to-nearest bf16 pieces hold the 24-bit significand), and
//   a*b ~= a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 with every product exact and the sum in fp32 (dropped terms <= 2^-23 |ab|).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_split_lab.hip -o build/gemm_split_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 p = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, p);
}
// two floats -> three packed bf16 pairs (lo half = a, hi half = b)
__device__ __forceinline__ void split2(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16); rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
}

// Wp layout: [N][K/8][3][8] bf16.  VAR bits: 1 = three products only (a0b0+a0b1+a1b0)
template <int BM, int BN, int BK, int WM, int WN, int VAR>
__global__ __launch_bounds__(WM * WN * 64) void gemm_split(const float* __restrict__ A, const uint16_t* __restrict__ Wp, float* __restrict__ C,
                                                            int M, int N, int K) {
    constexpr int NT = WM * WN * 64, KG = BK / 8, ROWB = KG * 48 + 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM * (BK / 4) / NT;            // float4 units per thread (A)
    constexpr int UB = BN * KG * 3, RB = (UB + NT - 1) / NT;   // 16-byte units per thread (B)
    static_assert(BM * (BK / 4) % NT == 0, "tile/threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 2 * BM * ROWB;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM, nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tile_m = wgid / tiles_n, tile_n = wgid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    f32x4 ra[RA];
    u32x4 rb[RB];
    const size_t wrow = (size_t)(K / 8) * 48;         // bytes per W row
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; i++) {
            const int u = tid + NT * i, row = u / (BK / 4), c4 = u % (BK / 4);
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * K + k0 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int u = tid + NT * i, row = u / (KG * 3), c = u % (KG * 3);
            if (UB % NT == 0 || u < UB)
                rb[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Wp) + (size_t)(n0 + row) * wrow + (size_t)(k0 / 8) * 48 + c * 16);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; i++) {
            const int u = tid + NT * i, row = u / (BK / 4), c4 = u % (BK / 4);
            char* dst = As + (buf * BM + row) * ROWB + (c4 >> 1) * 48 + (c4 & 1) * 8;
            uint32_t a0, a1, a2, b0, b1, b2;
            split2(ra[i][0], ra[i][1], a0, a1, a2);
            split2(ra[i][2], ra[i][3], b0, b1, b2);
            *reinterpret_cast<u32x2*>(dst) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(dst + 16) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(dst + 32) = u32x2{a2, b2};
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int u = tid + NT * i, row = u / (KG * 3), c = u % (KG * 3);
            if (UB % NT == 0 || u < UB) *reinterpret_cast<u32x4*>(Bs + (buf * BN + row) * ROWB + c * 16) = rb[i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int nk = K / BK;
    load_regs(0);
    store_lds(0);
    if (nk > 1) load_regs(BK);
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) store_lds(buf ^ 1);
        if (kt + 2 < nk) load_regs((kt + 2) * BK);
        const char* Ab = As + (buf * BM + wm * (BM / WM) + frow) * ROWB + fh * 48;
        const char* Bb = Bs + (buf * BN + wn * (BN / WN) + frow) * ROWB + fh * 48;
#pragma unroll
        for (int s = 0; s < BK / 16; s++) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int p = 0; p < 3; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * ROWB + s * 96 + p * 16);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int p = 0; p < 3; p++) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * ROWB + s * 96 + p * 16);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (!(VAR & 1)) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    const int half = lane >> 5, lcol = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = n0 + wn * (BN / WN) + j * 32 + lcol;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][e];
            }
    }
}

static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
    return (uint16_t)u;
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int BM, int BN, int BK, int WM, int WN, int VAR>
double run(const char* name, const float* A, const uint16_t* Wp, float* C, int M, int N, int K, int iters) {
    auto kern = gemm_split<BM, BN, BK, WM, WN, VAR>;
    const int lds = 2 * (BM + BN) * ((BK / 8) * 48 + 16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (M % BM || N % BN || K % BK) { printf("%-36s skipped (shape)\n", name); return 0; }
    const int blocks = (M / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, A, Wp, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, 0, A, Wp, C, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-36s M=%5d N=%5d K=%5d blocks=%5d lds=%6d  %8.1f us  %7.2f TF(fp32-equivalent)\n", name, M, N, K, blocks, lds, us, tf);
    return us;
}

int main() {
    const int shapes[][3] = {{4096, 4096, 4096}, {18432, 4096, 1024}, {18432, 3072, 1024}, {18432, 1024, 1024}, {18432, 1024, 4096},
                             {18432, 768, 768}, {18432, 768, 3072}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() * 0.05f;
        std::vector<uint16_t> hWp((size_t)N * K * 3);
        for (int n = 0; n < N; n++)
            for (int k = 0; k < K; k++) {
                float x = hW[(size_t)n * K + k];
                uint16_t p0 = bf16_rne(x); float r1 = x - bf16_f(p0);
                uint16_t p1 = bf16_rne(r1); float r2 = r1 - bf16_f(p1);
                uint16_t p2 = bf16_rne(r2);
                size_t base = ((size_t)n * (K / 8) + k / 8) * 24 + (k % 8);
                hWp[base] = p0; hWp[base + 8] = p1; hWp[base + 16] = p2;
            }
        float *A, *C; uint16_t* Wp;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Wp, hWp.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
        const int it = 20;
        auto check = [&](const char* what) {
            std::vector<float> hC((size_t)8 * N);
            CK(hipMemcpy(hC.data(), C + (size_t)(M - 8) * N, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxrel = 0, sumabs = 0;
            for (int rr = 0; rr < 8; rr++)
                for (int n = 0; n < N; n += 7) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; k++) { double p = (double)hA[(size_t)(M - 8 + rr) * K + k] * (double)hW[(size_t)n * K + k]; ref += p; mag += fabs(p); }
                    double e = fabs((double)hC[(size_t)rr * N + n] - ref) / mag;
                    if (e > maxrel) maxrel = e;
                    sumabs += e;
                }
            printf("   %s: max |err| / sum|a||b| = %.3e\n", what, maxrel);
        };
        run<128, 128, 32, 2, 2, 0>("128x128x32 4w six", A, Wp, C, M, N, K, it); check("six products");
        run<128, 128, 32, 2, 2, 1>("128x128x32 4w three", A, Wp, C, M, N, K, it); check("three products");
        run<128, 128, 16, 2, 2, 0>("128x128x16 4w six", A, Wp, C, M, N, K, it);
        run<128, 128, 32, 4, 2, 0>("128x128x32 8w(4x2) six", A, Wp, C, M, N, K, it);
        run<128, 128, 32, 2, 4, 0>("128x128x32 8w(2x4) six", A, Wp, C, M, N, K, it);
        run<256, 128, 16, 4, 2, 0>("256x128x16 8w six", A, Wp, C, M, N, K, it);
        run<128, 64, 32, 2, 2, 0>("128x64x32 4w six", A, Wp, C, M, N, K, it);
        run<64, 64, 32, 2, 2, 0>("64x64x32 4w six", A, Wp, C, M, N, K, it);
        CK(hipFree(A)); CK(hipFree(Wp)); CK(hipFree(C));
    }
    return 0;
}
