// Developer lab (not part of the product): variants of the fp32-MFMA GEMM main loop, timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o build/gemm_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// VARIANT bits: 1 = setprio around MFMA, 2 = no bounds predication, 4 = preload all fragments of the slab,
// 8 = direct-to-LDS (glds) staging with XOR swizzle (implies no register staging)
template <int BM, int BN, int WM, int WN, int VAR>
__global__ __launch_bounds__(WM * WN * 64) void gemm_lab(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                          int M, int N, int K) {
    constexpr int BK = 32, LDP = 36, NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;      // MFMA tiles per wave
    constexpr int RA = BM * 8 / NT, RB = BN * 8 / NT;        // float4 per thread per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + 2 * BM * LDP;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM, nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tile_m = wgid / tiles_n, tile_n = wgid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    constexpr int ROWS_PER_PASS = NT / 8;
    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; i++) {
            const int gm = m0 + lrow + ROWS_PER_PASS * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((VAR & 2) || gm < M) v = *reinterpret_cast<const f32x4*>(A + (size_t)gm * K + k0 + lc4);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int gn = n0 + lrow + ROWS_PER_PASS * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((VAR & 2) || gn < N) v = *reinterpret_cast<const f32x4*>(W + (size_t)gn * K + k0 + lc4);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; i++) *reinterpret_cast<f32x4*>(As + (buf * BM + lrow + ROWS_PER_PASS * i) * LDP + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; i++) *reinterpret_cast<f32x4*>(Bs + (buf * BN + lrow + ROWS_PER_PASS * i) * LDP + lc4) = rb[i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int nk = K / BK;
    load_tile(0);
    store_tile(0);
    if (VAR & 16) { if (nk > 1) load_tile(BK); }
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    if (VAR & 16) {
        // software pipeline: at the top of iteration kt the registers hold slab kt+1 (loaded during kt-1); write it to
        // the other LDS buffer right away (that buffer was last read in kt-1, all waves passed the barrier since),
        // then reuse the registers for slab kt+2.  The end of the iteration is a bare barrier.
        for (int kt = 0; kt < nk; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nk) store_tile(buf ^ 1);
            if (kt + 2 < nk) load_tile((kt + 2) * BK);
            const float* Ab = As + (buf * BM + wm * (BM / WM) + frow) * LDP + fk;
            const float* Bb = Bs + (buf * BN + wn * (BN / WN) + frow) * LDP + fk;
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
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
            __syncthreads();
        }
    } else
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* Ab = As + (buf * BM + wm * (BM / WM) + frow) * LDP + fk;
        const float* Bb = Bs + (buf * BN + wn * (BN / WN) + frow) * LDP + fk;
        if (VAR & 4) {
            f32x4 af[4][TM], bf[4][TN];
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[kb][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDP + kb * 8);
#pragma unroll
                for (int j = 0; j < TN; j++) bf[kb][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDP + kb * 8);
            }
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kb = 0; kb < 4; kb++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb][i][t], bf[kb][j][t], acc[i][j], 0, 0, 0);
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
        } else {
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
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
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
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

template <int BM, int BN, int WM, int WN, int VAR>
double run(const char* name, const float* A, const float* W, float* C, int M, int N, int K, int iters) {
    auto kern = gemm_lab<BM, BN, WM, WN, VAR>;
    const int lds = 2 * (BM + BN) * 36 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, 0, A, W, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, 0, A, W, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms / iters * 1e-3, tf = 2.0 * M * N * K / t / 1e12;
    printf("%-34s M=%5d N=%5d K=%5d blocks=%5d  %8.1f us  %7.2f TF\n", name, M, N, K, nwg, t * 1e6, tf);
    return tf;
}

int main() {
    const int shapes[][3] = {{4096, 4096, 4096}, {18432, 4096, 1024}, {18432, 3072, 1024}, {18432, 1024, 1024}, {18432, 1024, 4096},
                             {18432, 768, 768}, {18432, 768, 3072}};
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s[0] * s[2]); maxW = std::max(maxW, (size_t)s[1] * s[2]); maxC = std::max(maxC, (size_t)s[0] * s[1]);
    }
    float *A, *W, *C, *Cref;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&Cref, maxC * 4));
    std::vector<float> h(std::max(maxA, maxW));
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(W, h.data(), maxW * 4, hipMemcpyHostToDevice));
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2], it = 10;
        run<128, 128, 2, 2, 0>("128x128 4w base", A, W, Cref, M, N, K, it);
        run<128, 128, 2, 2, 1>("128x128 4w prio", A, W, C, M, N, K, it);
        run<128, 128, 2, 2, 2>("128x128 4w nopred", A, W, C, M, N, K, it);
        run<128, 128, 2, 2, 18>("128x128 4w nopred early-write pipe", A, W, C, M, N, K, it);
        run<128, 128, 2, 2, 19>("128x128 4w nopred pipe+prio", A, W, C, M, N, K, it);
        run<256, 128, 4, 2, 18>("256x128 8w nopred pipe", A, W, C, M, N, K, it);
        run<128, 64, 2, 2, 2>("128x64 4w nopred", A, W, C, M, N, K, it);
        run<128, 64, 2, 2, 18>("128x64 4w nopred pipe", A, W, C, M, N, K, it);
        run<64, 64, 2, 2, 2>("64x64 4w nopred", A, W, C, M, N, K, it);
        run<64, 64, 2, 2, 18>("64x64 4w nopred pipe", A, W, C, M, N, K, it);
        // correctness of the last variant vs base on a few entries
        std::vector<float> c0(64), c1(64);
        CK(hipMemcpy(c0.data(), Cref + (size_t)(M - 1) * N + N - 64, 256, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c1.data(), C + (size_t)(M - 1) * N + N - 64, 256, hipMemcpyDeviceToHost));
        double d = 0;
        for (int i = 0; i < 64; i++) d = std::max(d, (double)fabsf(c0[i] - c1[i]));
        printf("   maxdiff(last row tail) = %g\n", d);
    }
    return 0;
}
