// Lab: what read rate does the aligner's access pattern reach without its arithmetic?  (tools/, not part of liba3r)
//   A: the main kernel's pattern -- a workgroup of 256 threads owns 1024 pixels of image n and walks deg edge sides, each 3 x 16 B of
//      points + 16 B of weights per thread from two arrays at edge-dependent offsets (two register buffers);
//   B: the same bytes from ONE packed stream [n][chunk][k][thread][64 B];
//   C: like A but the minimum arithmetic replaced by the real kernel's ~60 VALU per pixel side (fma chain).
// hipcc --offload-arch=gfx950 -O3 tools/align_stream_lab.hip -o build/lab/align_stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256, CHUNK = 1024;
template <int MODE, int WORK>
__global__ __launch_bounds__(TPB, 4) void k(const float* __restrict__ pred, const float* __restrict__ wt, const f32x4* __restrict__ packed,
                                            const int* __restrict__ inc, int deg, int P, float* out) {
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int pix0 = chunk * CHUNK + tid * 4;
    f32x4 acc = {0, 0, 0, 0};
    auto load = [&](int kk, f32x4& a, f32x4& b, f32x4& c, f32x4& w) {
        if (MODE == 1) {
            const f32x4* p = packed + ((((size_t)n * gridDim.x + chunk) * deg + kk) * TPB + tid) * 4;
            a = p[0]; b = p[1]; c = p[2]; w = p[3];
        } else {
            const int e = __builtin_amdgcn_readfirstlane(inc[n * deg + kk]);
            const f32x4* xp = reinterpret_cast<const f32x4*>(pred + ((size_t)e * P + pix0) * 3);
            a = xp[0]; b = xp[1]; c = xp[2];
            w = *reinterpret_cast<const f32x4*>(wt + (size_t)e * P + pix0);
        }
    };
    f32x4 a0, b0, c0, w0, a1, b1, c1, w1;
    load(0, a0, b0, c0, w0);
    auto use = [&](f32x4 a, f32x4 b, f32x4 c, f32x4 w) {
        f32x4 t = a * w + b;
#pragma unroll
        for (int i = 0; i < WORK; i++) t = t * c + a;
        acc += t;
    };
#pragma unroll 1
    for (int kk = 0; kk < deg; kk += 2) {
        if (kk + 1 < deg) load(kk + 1, a1, b1, c1, w1);
        use(a0, b0, c0, w0);
        if (kk + 1 < deg) {
            if (kk + 2 < deg) load(kk + 2, a0, b0, c0, w0);
            use(a1, b1, c1, w1);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    const int N = 16, E = 84, P = 196608, deg = 2 * 2 * E / N / 2 * 1;   // edge sides per image: 2 E sides over N images... = 10.5 -> use 21 for symmetrised
    const int DEG = 21, EE = 2 * E;                                     // 168 (edge, side) arrays
    float *pred, *wt, *out; f32x4* packed; int* inc;
    hipMalloc(&pred, (size_t)EE * P * 12); hipMalloc(&wt, (size_t)EE * P * 4); hipMalloc(&packed, (size_t)N * DEG * P * 16); hipMalloc(&out, 64);
    hipMemset(pred, 0, (size_t)EE * P * 12); hipMemset(wt, 0, (size_t)EE * P * 4); hipMemset(packed, 0, (size_t)N * DEG * P * 16);
    std::vector<int> h(N * DEG);
    for (int n = 0; n < N; n++) for (int kk = 0; kk < DEG; kk++) h[n * DEG + kk] = (n * 37 + kk * 11) % EE;
    hipMalloc(&inc, h.size() * 4); hipMemcpy(inc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    dim3 grid(P / CHUNK, N);
    const double bytes = (double)N * DEG * P * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(TPB), 0, 0, pred, wt, packed, inc, DEG, P, out);
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, grid, dim3(TPB), 0, 0, pred, wt, packed, inc, DEG, P, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
    };
    run("A scattered two-array, no work", k<0, 0>);
    run("B packed one stream, no work", k<1, 0>);
    run("A scattered, 8 dependent pk-fma", k<0, 8>);
    run("B packed, 8 dependent pk-fma", k<1, 8>);
    run("A scattered, 16 fma", k<0, 16>);
    run("B packed, 16 fma", k<1, 16>);
    return 0;
}
