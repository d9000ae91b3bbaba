// Per-pixel passes of the aligner's construction and of the MST initialisation, as native launches.
//
// Round 2 measured one 16-frame clip at 1.53 s of which 1.0 s was host glue: the confidences travelled to the host and back
// (132 MB each way), the per-image confidence maximum was a Python loop of CPU maximums, and every torch elementwise op the
// initialisation touched for the first time paid a code-object load (10-120 ms each: square, sign, norm, nan_to_num, matmul ->
// rocBLAS start-up ...).  These kernels replace those passes; the small algebra between them (a handful of 3x4 matrices and
// quaternions per image / edge) is done in numpy on the host from ONE read-back.  Reference semantics, all under /root/reference:
//   a3r_conf_prepare      conf_trf + the per-edge mean confidence     dust3r/cloud_opt/base_opt.py:133-140, commons.py:20-25,42-55
//   a3r_im_conf_max       per-image confidence = max over its edges   dust3r/cloud_opt/base_opt.py:169-175
//   a3r_weiszfeld_focal   estimate_focal_knowing_depth('weiszfeld')   dust3r/post_process.py:36-60
//   a3r_sim3_apply        geotrf(sRT_to_4x4(s, R, T), pts)            dust3r/cloud_opt/init_im_poses.py:226-233,415-418
//   a3r_depth_init        _set_depthmap of init_from_pts3d           dust3r/cloud_opt/init_im_poses.py:116-126, optimizer.py:131-135
//   a3r_mask_gt           conf > min_conf_thr                        dust3r/cloud_opt/init_im_poses.py:239-241
// Reductions are fixed-order (strided per-thread partial sums in float64, then an LDS tree): bitwise reproducible.
#include "common.h"
#include <cfloat>
#include <cmath>

namespace a3r {

// sum of v over the 1024 threads of a block (every thread gets the result); red: 1024 doubles of LDS
__device__ __forceinline__ double block_sum_1024(double v, double* red) {
    const int t = threadIdx.x;
    __syncthreads();                       // red may still be read by the previous call's consumers
    red[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    return red[0];
}

__device__ __forceinline__ float conf_trf(float c, int mode) {          // commons.py:42-55
    return mode == 0 ? logf(c) : mode == 1 ? sqrtf(c) : mode == 2 ? c - 1.f : c;
}

// blockIdx.x = 2 e + side.  w = trf(conf) (skipped when w_* is null), mean[2 e + side] = mean(conf) in float64 -> float
__global__ __launch_bounds__(1024) void conf_prepare_kernel(const float* __restrict__ conf_i, const float* __restrict__ conf_j, long P,
                                                            int mode, float* __restrict__ w_i, float* __restrict__ w_j,
                                                            float* __restrict__ mean) {
    __shared__ double red[1024];
    const int e = blockIdx.x >> 1, side = blockIdx.x & 1;
    const float* c = (side ? conf_j : conf_i) + (size_t)e * P;
    float* w = side ? w_j : w_i;
    if (w) w += (size_t)e * P;
    double s = 0.0;
    const long P4 = P >> 2;
    for (long k = threadIdx.x; k < P4; k += 1024) {
        const f32x4 v = reinterpret_cast<const f32x4*>(c)[k];
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        if (w) reinterpret_cast<f32x4*>(w)[k] = f32x4{conf_trf(v.x, mode), conf_trf(v.y, mode), conf_trf(v.z, mode), conf_trf(v.w, mode)};
    }
    for (long k = P4 * 4 + threadIdx.x; k < P; k += 1024) {
        s += (double)c[k];
        if (w) w[k] = conf_trf(c[k], mode);
    }
    const double tot = block_sum_1024(s, red);
    if (threadIdx.x == 0 && mean) mean[blockIdx.x] = (float)(tot / (double)P);
}

// out[n, p] = max(0, max over edges e with ei[e] == n of conf_i[e, p], max over edges with ej[e] == n of conf_j[e, p])
__global__ __launch_bounds__(256) void im_conf_max_kernel(const float* __restrict__ conf_i, const float* __restrict__ conf_j,
                                                          const int* __restrict__ ei, const int* __restrict__ ej, int E, long P,
                                                          float* __restrict__ out) {
    const int n = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float m = 0.f;                                                   // base_opt.py:170: torch.zeros(hw)
    for (int e = 0; e < E; e++) {                                    // (ei / ej reads are wave-uniform: scalar loads)
        if (ei[e] == n) m = fmaxf(m, conf_i[(size_t)e * P + p]);
        if (ej[e] == n) m = fmaxf(m, conf_j[(size_t)e * P + p]);
    }
    out[(size_t)n * P + p] = m;
}

// one block per point map [H, W, 3]; principal point at the image centre (post_process.py:41-43 with pp = (W/2, H/2))
__global__ __launch_bounds__(1024) void weiszfeld_focal_kernel(const float* __restrict__ pts, int H, int W, int iters, float* __restrict__ focal) {
    __shared__ double red[1024];
    const float* p = pts + (size_t)blockIdx.x * H * W * 3;
    const long P = (long)H * W;
    const float cx = W / 2.f, cy = H / 2.f;
    // per pixel: u = (x - cx, y - cy), q = xy / z with inf / nan -> 0 (post_process.py:45 nan_to_num(posinf=0, neginf=0))
    auto load = [&](long k, float& ux, float& uy, float& qx, float& qy) {
        const int y = (int)(k / W), x = (int)(k - (long)y * W);
        ux = (float)x - cx; uy = (float)y - cy;
        const float z = p[3 * k + 2];
        qx = p[3 * k] / z; qy = p[3 * k + 1] / z;
        if (!(fabsf(qx) <= FLT_MAX)) qx = 0.f;
        if (!(fabsf(qy) <= FLT_MAX)) qy = 0.f;
    };
    double a = 0.0, b = 0.0;
    for (long k = threadIdx.x; k < P; k += 1024) {
        float ux, uy, qx, qy;
        load(k, ux, uy, qx, qy);
        a += (double)(qx * ux + qy * uy);
        b += (double)(qx * qx + qy * qy);
    }
    a = block_sum_1024(a, red);
    b = block_sum_1024(b, red);
    float f = (float)(a / b);                                          // post_process.py:50-51 (the means' 1/P cancels)
    for (int it = 0; it < iters; it++) {                               // post_process.py:54-59
        double wa = 0.0, wb = 0.0;
        for (long k = threadIdx.x; k < P; k += 1024) {
            float ux, uy, qx, qy;
            load(k, ux, uy, qx, qy);
            const float dx = ux - f * qx, dy = uy - f * qy;
            const float w = 1.f / fmaxf(sqrtf(dx * dx + dy * dy), 1e-8f);
            wa += (double)(w * (qx * ux + qy * uy));
            wb += (double)(w * (qx * qx + qy * qy));
        }
        wa = block_sum_1024(wa, red);
        wb = block_sum_1024(wb, red);
        f = (float)(wa / wb);
    }
    if (threadIdx.x == 0) focal[blockIdx.x] = fmaxf(f, 0.f);           // post_process.py:60 focal.clip(min=min_focal = 0)
}

// y = k R x + T with (s, R row-major, T) = sol[0..12] read from DEVICE memory (no host round trip between the registration that
// produced it and this use); k = s when with_scale, else 1.  post: every output additionally multiplied by post (1 = none)
__global__ __launch_bounds__(256) void sim3_apply_kernel(const float* __restrict__ x, const float* __restrict__ sol, int with_scale,
                                                         float post, float* __restrict__ y, long P) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float k = with_scale ? sol[0] : 1.f;
    const float a = x[3 * p], b = x[3 * p + 1], c = x[3 * p + 2];
#pragma unroll
    for (int r = 0; r < 3; r++)
        y[3 * p + r] = ((sol[1 + 3 * r] * a + sol[2 + 3 * r] * b + sol[3 + 3 * r] * c) * k + sol[10 + r]) * post;
}

// depth[n, p] = log(z) with z the camera-space depth of world point pts[n, p] under the world-to-camera rows w2c[n] = [R | t] (3x4,
// row-major, DEVICE), pts scaled by `scale` first; log(z <= 0 or nan) -> 0, log(+inf) -> FLT_MAX  (= .log().nan_to_num(neginf=0))
__global__ __launch_bounds__(256) void depth_init_kernel(const float* __restrict__ pts, const float* __restrict__ w2c, float scale, long P,
                                                         float* __restrict__ depth) {
    const int n = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* q = pts + ((size_t)n * P + p) * 3;
    const float* m = w2c + n * 12 + 8;
    const float z = m[0] * (q[0] * scale) + m[1] * (q[1] * scale) + m[2] * (q[2] * scale) + m[3];
    float d = 0.f;
    if (z > 0.f) d = z > FLT_MAX ? FLT_MAX : logf(z);
    depth[(size_t)n * P + p] = d;
}

__global__ __launch_bounds__(256) void mask_gt_kernel(const float* __restrict__ x, float thr, unsigned char* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[i] > thr ? 1 : 0;
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_conf_prepare(const float* conf_i, const float* conf_j, int E, long P, int mode, float* w_i, float* w_j,
                                float* edge_mean, void* stream) {
    A3R_CHECK_ARG(conf_i && conf_j && E > 0 && P > 0, "a3r_conf_prepare: bad argument");
    A3R_CHECK_ARG(mode >= 0 && mode <= 3, "a3r_conf_prepare: mode must be 0 (log), 1 (sqrt), 2 (m1) or 3 (id)");
    A3R_CHECK_ARG((w_i == nullptr) == (w_j == nullptr), "a3r_conf_prepare: w_i and w_j come together");
    A3R_CHECK_ARG(((reinterpret_cast<uintptr_t>(conf_i) | reinterpret_cast<uintptr_t>(conf_j) | reinterpret_cast<uintptr_t>(w_i) |
                    reinterpret_cast<uintptr_t>(w_j)) & 15) == 0 && P % 4 == 0, "a3r_conf_prepare: 16-byte aligned maps with P %% 4 == 0 required");
    hipLaunchKernelGGL(conf_prepare_kernel, dim3(2 * E), dim3(1024), 0, as_stream(stream), conf_i, conf_j, P, mode, w_i, w_j, edge_mean);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_im_conf_max(const float* conf_i, const float* conf_j, const int* ei, const int* ej, int E, int N, long P, float* out,
                               void* stream) {
    A3R_CHECK_ARG(conf_i && conf_j && ei && ej && out && E > 0 && N > 0 && N <= 65535 && P > 0, "a3r_im_conf_max: bad argument");
    hipLaunchKernelGGL(im_conf_max_kernel, dim3((unsigned)((P + 255) / 256), N), dim3(256), 0, as_stream(stream), conf_i, conf_j, ei, ej, E, P, out);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_weiszfeld_focal(const float* pts3d, int B, int H, int W, int iterations, float* focal, void* stream) {
    A3R_CHECK_ARG(pts3d && focal && B > 0 && H > 0 && W > 0 && iterations >= 0, "a3r_weiszfeld_focal: bad argument");
    hipLaunchKernelGGL(weiszfeld_focal_kernel, dim3(B), dim3(1024), 0, as_stream(stream), pts3d, H, W, iterations, focal);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_sim3_apply(const float* x, const float* sol, int with_scale, float post, float* y, long P, void* stream) {
    A3R_CHECK_ARG(x && sol && y && P > 0, "a3r_sim3_apply: bad argument");
    hipLaunchKernelGGL(sim3_apply_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), x, sol, with_scale, post, y, P);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_depth_init(const float* pts, const float* w2c, float scale, int N, long P, float* depth, void* stream) {
    A3R_CHECK_ARG(pts && w2c && depth && N > 0 && N <= 65535 && P > 0, "a3r_depth_init: bad argument");
    hipLaunchKernelGGL(depth_init_kernel, dim3((unsigned)((P + 255) / 256), N), dim3(256), 0, as_stream(stream), pts, w2c, scale, P, depth);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_mask_gt(const float* x, float thr, unsigned char* out, long n, void* stream) {
    A3R_CHECK_ARG(x && out && n > 0, "a3r_mask_gt: bad argument");
    hipLaunchKernelGGL(mask_gt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, thr, out, n);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
