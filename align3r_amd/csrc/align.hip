// Global-alignment inner loop for gfx950: fused loss + analytic gradients + Adam.
//
// Replaces one global_alignment_iter of the reference (dust3r/cloud_opt/base_opt.py:450-464):
//   loss = PointCloudOptimizer.forward()   dust3r/cloud_opt/optimizer.py:223-241
//   loss.backward()                        (autograd in the reference; hand-derived here)
//   torch.optim.Adam(betas=(.9,.9)).step() base_opt.py:435
//
// Design (HBM-bound streaming + reductions; no GEMM shape anywhere):
//   * image-major: a workgroup owns 1024 pixels of one image and walks the (edge, side) pairs
//     incident to that image.  The projected point of each pixel is computed once and kept in
//     registers, the per-pixel depth gradient is complete when the walk ends, so the Adam update of
//     the per-pixel parameter happens in the same kernel: no gradient map ever touches HBM.
//     Traffic per iteration = 32*E*P (pred 12 B + weight 4 B, both sides) + 24*N*P (param, m, v r/w).
//   * the 12+1 per-edge sums (sum g (x) X, sum g, loss) are reduced with DPP inside 16-lane rows, then
//     through LDS across the workgroup, and written as per-chunk partials that a second, tiny kernel
//     adds in a fixed order: results are bitwise reproducible (no float atomics).
//   * the pose / focal / scale parameters (a few KB) get their chain rule + Adam in two small kernels.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <new>

namespace a3r {

constexpr int PXT = 4;                 // pixels per thread
constexpr int TPB = 256;
constexpr int CHUNK = PXT * TPB;       // pixels per workgroup
// register buffers of raw edge data per thread: 2 (one edge side in flight behind the one being consumed; fits four waves per SIMD)
// or 3 (two in flight; lab variant, needs A3R_ALIGN_MIN_WAVES=3)
#ifndef A3R_ALIGN_NBUF
#define A3R_ALIGN_NBUF 2
#endif
constexpr int EB = A3R_ALIGN_NBUF == 3 ? 9 : 8;   // (edge,side) entries per LDS reduction batch (a multiple of the buffers)
constexpr int MAX_INC = 2048;          // edge sides incident to one image (their codes sit in LDS: 8 KB)
constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.9f, ADAM_EPS = 1e-8f;  // base_opt.py:435

struct AlignDev {
    int E, N, P, nchunks;
    int norm_pw_scale, train_poses, train_focals, train_pp, train_adaptors;
    float base_scale, pw_break, focal_break;
    float inv_area_i, inv_area_j;
    const float *pred_i, *pred_j, *w_i, *w_j, *mono, *pp0;
    float *pw_poses, *pw_adaptors, *depth, *shifts, *im_poses, *im_focals, *im_pp;
    float *adam_pw_poses, *adam_depth, *adam_small, *adam_pw_adaptors;
    // workspace
    float *edge_xf, *img_xf, *partE, *partN, *gE, *gN, *lossE, *gA;
    float *sumE, *sumN;               // [2E][16] per incidence slot, [N][16] per image: chunk partials added in a fixed order
    int* tick;                        // [N + 1] arrival counters of the in-kernel tail: per image, then all images (zero between launches)
    int fused_tail;                   // 1: the main kernel finishes the iteration itself (last-block-done tickets); 0: finalize A/B launches
    const int *inc_ptr, *inc, *slot_of, *imw, *imarea, *order;
    float* loss_history;
    // cloud_opt_flow extras (a3r_align_set_flow); all zero / null for the plain cloud_opt aligner
    int shared_focal;
    float tsw, trans_w;               // temporal smoothing weight, translation weight (optimizer.py:516-519,559-572)
    float flow_w, flow_thre, pxl_thre;
    int flow_on;                      // the ego-flow term takes part in THIS launch (set by the host per iteration)
    int fH, fW;
    const float *flow_ij, *flow_ji;   // [E, 2, P]
    const unsigned char* dyn;         // [N, P] 1 = dynamic pixel (excluded)
    const int* other;                 // [2E] target image of incidence slot k
    float *gflow;                     // [2, N, P, 3] unscaled d(sum loss)/d(world point) per direction
    float *partF, *sumF;              // [2E, nchunks, 20], [2E, 20]
    float *flow_state;                // [8]: c0, c1, flow_loss, dropped_now, dropped_sticky
    float *lossN;                     // [N] temporal-smoothing loss of the pair (n, n+1)
    // depth prior of cloud_opt_flow (a3r_align_set_depth_prior); weight 0 / null otherwise
    float prior_w;
    const float* prior_init;          // [N, P] log-depth parameters at the time of _set_init_depthmap
    const unsigned char* prior_dyn;   // [N, P] 1 = dynamic pixel (weight 2), may be null (all weights 1)
    float *gprior, *lossP;            // [N, P] d(weighted prior)/d(log-depth parameter); [N] per-image prior loss
};

struct AdamArgs {
    float lr, step_size, bc2_sqrt;
    int step;       // 0-based index into loss_history
};

// ------------------------------------------------------------------------------------------- math helpers
__device__ __forceinline__ void quat_to_R(const float* q, float* R, float* qn, float* nrm) {
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    float twx = tx * w, twy = ty * w, twz = tz * w;
    float txx = tx * x, txy = ty * x, txz = tz * x;
    float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
    if (qn) { qn[0] = x; qn[1] = y; qn[2] = z; qn[3] = w; }
    if (nrm) *nrm = n;
}

__device__ __forceinline__ void quat_backward(const float* qn, float nrm, const double* G, double* gq) {
    double x = qn[0], y = qn[1], z = qn[2], w = qn[3];
    double gx = 2 * (y * G[1] + z * G[2] + y * G[3] - 2 * x * G[4] - w * G[5] + z * G[6] + w * G[7] - 2 * x * G[8]);
    double gy = 2 * (-2 * y * G[0] + x * G[1] + w * G[2] + x * G[3] + z * G[5] - w * G[6] + z * G[7] - 2 * y * G[8]);
    double gz = 2 * (-2 * z * G[0] - w * G[1] + x * G[2] + w * G[3] - 2 * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
    double gw = 2 * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    double dot = gx * x + gy * y + gz * z + gw * w;
    gq[0] = (gx - dot * x) / nrm; gq[1] = (gy - dot * y) / nrm;
    gq[2] = (gz - dot * z) / nrm; gq[3] = (gw - dot * w) / nrm;
}

__device__ __forceinline__ float signed_expm1f(float x) {   // commons.py:118-120
    float s = (x > 0.f) - (x < 0.f);
    return s * expm1f(fabsf(x));
}
__device__ __forceinline__ float signed_expm1_grad(float x) { return x == 0.f ? 0.f : expf(fabsf(x)); }

// torch.optim.Adam single-tensor update (adam.py _single_tensor_adam), in place
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamArgs& a) {
    m = m + (g - m) * (1.f - ADAM_B1);
    v = v * ADAM_B2 + (1.f - ADAM_B2) * g * g;
    float denom = sqrtf(v) / a.bc2_sqrt + ADAM_EPS;
    p = p - a.step_size * (m / denom);
}

// ------------------------------------------------------------------------------------------- prep
// Per-edge [s*R*diag(a) | s*T], s, a and per-image [R | t], f, pp  (base_opt.py:177-229, optimizer.py:137-152)
__device__ void build_transforms(const AlignDev& d, float* sh /* >= TPB floats */) {
    const int tid = threadIdx.x;
    float part = 0.f;
    for (int e = tid; e < d.E; e += TPB) part += d.pw_poses[e * 8 + 7];
    part = wave_sum(part);
    if ((tid & 63) == 0) sh[tid >> 6] = part;
    __syncthreads();
    const float mean_ls = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)d.E;
    const float nf = d.norm_pw_scale ? expf(logf(d.base_scale) - mean_ls) : 1.f;
    for (int e = tid; e < d.E; e += TPB) {
        const float* p = d.pw_poses + e * 8;
        float R[9];
        quat_to_R(p, R, nullptr, nullptr);
        const float s = expf(p[7]) * nf;
        float ad[3] = {d.pw_adaptors[e * 2], d.pw_adaptors[e * 2], d.pw_adaptors[e * 2 + 1]};
        if (d.norm_pw_scale) {
            float m = (ad[0] + ad[1] + ad[2]) / 3.f;
            ad[0] -= m; ad[1] -= m; ad[2] -= m;
        }
        float a[3];
        for (int k = 0; k < 3; k++) a[k] = expf(ad[k] / d.pw_break);
        float* o = d.edge_xf + e * 16;
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) o[r * 4 + k] = s * R[r * 3 + k] * a[k];
            o[r * 4 + 3] = s * signed_expm1f(p[4 + r]);
        }
        o[12] = s; o[13] = a[0]; o[14] = a[1]; o[15] = a[2];
    }
    for (int n = tid; n < d.N; n += TPB) {
        const float* p = d.im_poses + n * 7;
        float R[9];
        quat_to_R(p, R, nullptr, nullptr);
        float* o = d.img_xf + n * 16;
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) o[r * 4 + k] = R[r * 3 + k];
            o[r * 4 + 3] = signed_expm1f(p[4 + r]);
        }
        o[12] = expf(d.im_focals[d.shared_focal ? 0 : n] / d.focal_break);
        o[13] = d.pp0[n * 2 + 0] + 10.f * d.im_pp[n * 2 + 0];
        o[14] = d.pp0[n * 2 + 1] + 10.f * d.im_pp[n * 2 + 1];
        o[15] = d.mono ? d.shifts[n] : 0.f;
    }
}

__global__ __launch_bounds__(TPB) void align_prep_kernel(AlignDev d) {
    __shared__ float sh[TPB];
    build_transforms(d, sh);
}

// ------------------------------------------------------------------------------------------- main
// grid (nchunks, N).  MODE 0: loss only; 1: gradients to g_depth (no update); 2: Adam update in place.
// VEC (P % 4 == 0): a thread owns 4 CONSECUTIVE pixels, so one edge-side is three 16-byte loads of points and
// one of weights per thread, and the next edge-side's loads are issued before the current one is consumed
// (two register buffers) -- the kernel is latency-bound otherwise (12 waves/CU x 4 KB in flight).
// !VEC: ragged P, pixels strided by the workgroup size, scalar loads.
// Raw loaded registers of one (edge, side); nothing may TOUCH them between the load and consume(), or the
// compiler has to wait for the data right where it was issued and the prefetch is gone.
template <bool VEC> struct EdgeData;
template <> struct EdgeData<true> { f32x4 a, b, c, w; };                    // 4 consecutive pixels: xyz xyz xyz xyz, wwww
template <> struct EdgeData<false> { float x[PXT][3]; float w[PXT]; };

// (Inline-asm loads with hand-placed counted waits were tried here to keep TWO edge sides per wave in flight -- hipcc's waitcnt
// pass puts a vmcnt wait in front of the next buffer's loads -- and abandoned: the register allocator copies the asm loads'
// destination registers at the loop's phi points while the data may still be landing, which no source-level form prevents.)
__device__ __forceinline__ void load_edge(const AlignDev& d, int code, int P, int pix0, const bool* valid, EdgeData<true>& o) {
    const int e = code >> 1, side = code & 1;
    const float* X = (side ? d.pred_j : d.pred_i) + (size_t)e * P * 3;
    const float* Wt = (side ? d.w_j : d.w_i) + (size_t)e * P;
    const int p = valid[0] ? pix0 : 0;     // the 4 pixels are valid together (P % 4 == 0)
    const f32x4* xp = reinterpret_cast<const f32x4*>(X + (size_t)p * 3);
    o.a = xp[0]; o.b = xp[1]; o.c = xp[2];
    o.w = *reinterpret_cast<const f32x4*>(Wt + p);
}
__device__ __forceinline__ void load_edge(const AlignDev& d, int code, int P, int pix0, const bool* valid, EdgeData<false>& o) {
    const int e = code >> 1, side = code & 1;
    const float* X = (side ? d.pred_j : d.pred_i) + (size_t)e * P * 3;
    const float* Wt = (side ? d.w_j : d.w_i) + (size_t)e * P;
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const size_t pp = valid[i] ? pix0 + i * TPB : 0;
        o.x[i][0] = X[pp * 3 + 0]; o.x[i][1] = X[pp * 3 + 1]; o.x[i][2] = X[pp * 3 + 2];
        o.w[i] = Wt[pp];
    }
}
__device__ __forceinline__ void unpack_edge(const EdgeData<true>& e, float (*x)[3], float* w) {
    x[0][0] = e.a.x; x[0][1] = e.a.y; x[0][2] = e.a.z;
    x[1][0] = e.a.w; x[1][1] = e.b.x; x[1][2] = e.b.y;
    x[2][0] = e.b.z; x[2][1] = e.b.w; x[2][2] = e.c.x;
    x[3][0] = e.c.y; x[3][1] = e.c.z; x[3][2] = e.c.w;
    w[0] = e.w.x; w[1] = e.w.y; w[2] = e.w.z; w[3] = e.w.w;
}
__device__ __forceinline__ void unpack_edge(const EdgeData<false>& e, float (*x)[3], float* w) {
#pragma unroll
    for (int i = 0; i < PXT; i++) { x[i][0] = e.x[i][0]; x[i][1] = e.x[i][1]; x[i][2] = e.x[i][2]; w[i] = e.w[i]; }
}

// the tail of an iteration (defined below the flow kernels)
struct TailOut { float *g_pw, *g_small, *loss_out, *g_adapt; };
__device__ void edge_chain(const AlignDev& d, int e, const double* s, bool loss_only);
__device__ void image_chain(const AlignDev& d, int n, double* s);
__device__ __forceinline__ f32x4 wave_sum_rows(const float* base, int nchunks, int lane);
template <int MODE>
__device__ void finalize_b_body(const AlignDev& d, const AdamArgs& ad, const TailOut& o, float* sh, double (*shd)[4]);

// Arrival of this workgroup on a ticket counter after its hand-off stores (MI355X guide, Guideline 16, write-through form).
// Producer side: EVERY handed-off byte is stored write-through at agent scope (store_wt below = global_store ... sc1), so no
// release fence -- which would write back the whole XCD L2, all the parameter / Adam lines of this iteration included, once per
// workgroup (measured: 280 us per iteration instead of 150) -- is needed: every storing wave drains its stores, the workgroup
// meets, ONE lane draws a ticket.  Consumer side: the workgroup that draws the last ticket acquires at agent scope (one
// buffer_inv, by one lane, then the barrier) before any of its threads reads what the others stored.
// Returns (to every thread) whether this workgroup is that last one.  `flag` is one int of LDS.
// 16 bytes per lane: a dword-wide sc1 store is one fabric write EACH (measured here: +26 us per iteration at E = 84, +120 us at
// E = 992 when the partial rows went out as 13 dword stores), a dwordx4 one costs what a plain store costs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(float* base /* wave-uniform */, unsigned off_bytes, f32x4 v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)off_bytes, 0, 16);   // aux 16 = sc1
}
__device__ __forceinline__ bool arrive_last(int* counter, int expected, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == expected - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// Lab build only (-DA3R_ALIGN_STAMPS, never the shipped library): s_memtime stamps of the phases of every workgroup of the main
// kernel, kept in SGPRs and stored once at the end (tools/align_stamps.py)
#ifdef A3R_ALIGN_STAMPS
__device__ unsigned long long g_align_stamps[8192 * 8];
#define A3R_STAMP(i) (stamp[i] = __builtin_amdgcn_s_memtime())
#else
#define A3R_STAMP(i)
#endif
// >= 4 waves per SIMD: the streaming loop needs <= 128 VGPRs; the (cold) tail may not raise the allocation
#ifndef A3R_ALIGN_MIN_WAVES
#define A3R_ALIGN_MIN_WAVES 4
#endif
template <bool MONO, bool L2, int MODE, bool VEC>
__global__ __launch_bounds__(TPB, VEC ? A3R_ALIGN_MIN_WAVES : 2) void align_main_kernel(
    AlignDev d, AdamArgs ad, float* g_depth, TailOut tout,
    // read-only, wave-uniform tables as noalias kernel arguments: the compiler can then use SCALAR loads
    // (s_load), which do not sit on the vector-memory counter -- with vector loads every lookup of the next
    // edge id drained the prefetched edge data (s_waitcnt vmcnt(0)) and serialised the loop
    const int* __restrict__ inc_ptr, const int* __restrict__ inc, const float* __restrict__ edge_xf,
    const float* __restrict__ img_xf, const int* __restrict__ imw, const int* __restrict__ imarea, const int* __restrict__ order) {
    __shared__ float red[2][EB][16][16];
#ifdef A3R_ALIGN_STAMPS
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
#endif
    A3R_STAMP(0);
    // images are dispatched longest first (order[] sorts them by their number of incident edge sides): the last round of
    // workgroups is then made of the short ones.  A dispatch slot's row of the table is {image, first and last incidence slot, the
    // first two (edge, side) codes}: ONE scalar load after which the first two edge sides are requested, before anything else --
    // the prologue used to be a chain of five dependent memory round trips (order -> inc_ptr -> inc -> LDS -> edge data) during
    // which the workgroup streamed nothing (15 % of its lifetime by s_memtime stamps, tools/align_stamps.py)
    const int* tb = order + blockIdx.y * 8;
    const int n = tb[0], kbeg = tb[1], kend = tb[2];
    const int chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = d.P;
    constexpr int PSTEP = VEC ? 1 : TPB;
    const int pix0 = chunk * CHUNK + (VEC ? tid * PXT : tid);       // pixel i of this thread: pix0 + i * PSTEP
    bool valid[PXT];
#pragma unroll
    for (int i = 0; i < PXT; i++) valid[i] = pix0 + i * PSTEP < P;
    EdgeData<VEC> ea, eb;
#if A3R_ALIGN_NBUF == 3
    EdgeData<VEC> ec;
#endif
    const float* ix = img_xf + n * 16;
    float R[9], T[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        R[r * 3 + 0] = ix[r * 4 + 0]; R[r * 3 + 1] = ix[r * 4 + 1]; R[r * 3 + 2] = ix[r * 4 + 2];
        T[r] = ix[r * 4 + 3];
    }
    const float f = ix[12], ppx = ix[13], ppy = ix[14], shift = ix[15];
    const int W = imw[n], area = imarea[n];
    const float invW = 1.f / (float)W, inv_f = 1.f / f;

    // forward of the image side: depth -> camera point -> world point (optimizer.py:190-200,244-251)
    auto pixel_forward = [&](int i, float rawv, float monov, float& dep, float& ddp, float& gxm, float& gym, float* rel) {
        const int p = pix0 + i * PSTEP;
        float gx = 0.f, gy = 0.f;
        if (p < area) {
            int y = (int)((float)p * invW);          // p < 2^24: exact up to +-1, fixed below
            int x = p - y * W;
            if (x < 0) { y--; x += W; }
            if (x >= W) { y++; x -= W; }
            gx = (float)x; gy = (float)y;
        }
        if (MONO) {
            const float es = expf(rawv);
            dep = monov * es + shift;
            ddp = monov * es;
        } else {
            dep = expf(rawv);
            ddp = dep;
        }
        gxm = gx - ppx; gym = gy - ppy;
        rel[0] = dep * gxm * inv_f;  // optimizer.py:251: depth * (pixel_grid - pp) / focal  (1/f: one IEEE divide per thread)
        rel[1] = dep * gym * inv_f;
        rel[2] = dep;
    };

    float raw[PXT], monov[PXT], proj[PXT][3], gp[PXT][3];
    if (VEC) {
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f}, m4 = {0.f, 0.f, 0.f, 0.f};
        if (valid[0]) {
            r4 = *reinterpret_cast<const f32x4*>(d.depth + (size_t)n * P + pix0);
            if (MONO) m4 = *reinterpret_cast<const f32x4*>(d.mono + (size_t)n * P + pix0);
        }
        // the first two edge sides are requested right behind the depth: the memory counter retires in order, so the forward
        // arithmetic below waits for the depth alone and runs while the edge data is still on its way
        if (kbeg < kend) load_edge(d, tb[3], P, pix0, valid, ea);
        if (kbeg + 1 < kend) load_edge(d, tb[4], P, pix0, valid, eb);
#if A3R_ALIGN_NBUF == 3
        if (kbeg + 2 < kend) load_edge(d, tb[5], P, pix0, valid, ec);
#endif
        raw[0] = r4.x; raw[1] = r4.y; raw[2] = r4.z; raw[3] = r4.w;
        monov[0] = m4.x; monov[1] = m4.y; monov[2] = m4.z; monov[3] = m4.w;
    } else {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const size_t off = (size_t)n * P + (valid[i] ? pix0 + i * PSTEP : 0);
            raw[i] = valid[i] ? d.depth[off] : 0.f;
            monov[i] = (MONO && valid[i]) ? d.mono[off] : 0.f;
        }
        if (kbeg < kend) load_edge(d, tb[3], P, pix0, valid, ea);
        if (kbeg + 1 < kend) load_edge(d, tb[4], P, pix0, valid, eb);
#if A3R_ALIGN_NBUF == 3
        if (kbeg + 2 < kend) load_edge(d, tb[5], P, pix0, valid, ec);
#endif
    }
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        float dep, ddp, gxm, gym, rel[3];
        pixel_forward(i, raw[i], monov[i], dep, ddp, gxm, gym, rel);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            proj[i][r] = R[r * 3] * rel[0] + R[r * 3 + 1] * rel[1] + R[r * 3 + 2] * rel[2] + T[r];
            gp[i][r] = 0.f;
        }
    }
    if (MODE != 0 && d.flow_on) {
        // ego-flow term (align_flow_kernel): its gradient w.r.t. this pixel's world point, scaled by weight / sum(mask)
        const float c0 = d.flow_state[0], c1 = d.flow_state[1];
        const size_t NP3 = (size_t)d.N * P * 3;
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            if (!valid[i]) continue;
            const float* gf = d.gflow + ((size_t)n * P + pix0 + i * PSTEP) * 3;
#pragma unroll
            for (int r = 0; r < 3; r++) gp[i][r] = c0 * gf[r] + c1 * gf[NP3 + r];
        }
    }

    // one (edge, side): residuals, loss, gradient w.r.t. the world point, per-edge sums (12 + loss)
    auto consume = [&](int code, const EdgeData<VEC>& ed, int buf, int kb) {
        float ex[PXT][3], ew[PXT];
        unpack_edge(ed, ex, ew);
        const int e = code >> 1, side = code & 1;
        const float* M = edge_xf + e * 16;
        const float m00 = M[0], m01 = M[1], m02 = M[2], m03 = M[3];
        const float m10 = M[4], m11 = M[5], m12 = M[6], m13 = M[7];
        const float m20 = M[8], m21 = M[9], m22 = M[10], m23 = M[11];
        const float inva = side ? d.inv_area_j : d.inv_area_i;
        float acc[13];
#pragma unroll
        for (int j = 0; j < 13; j++) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const float x0 = ex[i][0], x1 = ex[i][1], x2 = ex[i][2], w = valid[i] ? ew[i] : 0.f;
            const float r0 = proj[i][0] - (m00 * x0 + m01 * x1 + m02 * x2 + m03);
            const float r1 = proj[i][1] - (m10 * x0 + m11 * x1 + m12 * x2 + m13);
            const float r2 = proj[i][2] - (m20 * x0 + m21 * x1 + m22 * x2 + m23);
            const float sq = r0 * r0 + r1 * r1 + r2 * r2;
            float cf;
            if (L2) {
                acc[12] += sq * w * inva;
                cf = 2.f * w * inva;
            } else {
                // v_rsq_f32 (1 ulp) instead of an IEEE sqrt + an IEEE divide: this loop is VALU-bound (PMC:
                // 68 % VALU-active at 4 TB/s), and the two expansions were a quarter of its instructions
                const float inv = sq > 0.f ? __builtin_amdgcn_rsqf(sq) : 0.f;
                const float wa = w * inva;
                acc[12] += sq * inv * wa;
                cf = wa * inv;
            }
            if (MODE != 0) {
                const float g0 = cf * r0, g1 = cf * r1, g2 = cf * r2;
                gp[i][0] += g0; gp[i][1] += g1; gp[i][2] += g2;
                acc[0] += g0 * x0; acc[1] += g0 * x1; acc[2] += g0 * x2;
                acc[3] += g1 * x0; acc[4] += g1 * x1; acc[5] += g1 * x2;
                acc[6] += g2 * x0; acc[7] += g2 * x1; acc[8] += g2 * x2;
                acc[9] += g0; acc[10] += g1; acc[11] += g2;
            }
        }
        if (MODE == 0) {
            const float s = dpp_row_sum16(acc[12]);
            if ((lane & 15) == 0) red[buf][kb][wave * 4 + (lane >> 4)][12] = s;
        } else {
            // the 13 sums of a 16-lane row by a reduce-scatter (common.h): 29 VALU operations and ONE 16-byte LDS store per quad
            // instead of 13 four-step butterflies with a masked 4-byte store each -- this loop is bound by instruction issue
            // (tools/align_stream_lab.hip: its access pattern alone streams at 6.3 TB/s), and the butterflies, their DPP wait
            // states and the 13 exec-masked stores were a third of its instructions
            float v16[16], u[4];
#pragma unroll
            for (int j = 0; j < 13; j++) v16[j] = acc[j];
            v16[13] = v16[14] = v16[15] = 0.f;
            row_reduce_scatter16<13>(v16, u);
            if ((lane & 3) == 0) *reinterpret_cast<f32x4*>(&red[buf][kb][wave * 4 + (lane >> 4)][lane & 12]) = f32x4{u[0], u[1], u[2], u[3]};
        }
    };

    // The incidence codes are read with SCALAR loads (a uniform index into a noalias table: s_load, counted on lgkmcnt): a vector
    // load here would need s_waitcnt vmcnt(0) before its value could form the next address and would drain the edge data in flight.
    // (Rounds 1-2 copied the image's codes to LDS first, which cost the prologue a vector load, an LDS pass and a barrier.)
#ifndef A3R_ALIGN_SCALAR_CODES
#define A3R_ALIGN_SCALAR_CODES 1
#endif
#if A3R_ALIGN_SCALAR_CODES
    auto code_at = [&](int k) { return inc[__builtin_amdgcn_readfirstlane(k)]; };
#else
    __shared__ int s_inc[MAX_INC];
    for (int i = tid; i < kend - kbeg; i += TPB) s_inc[i] = inc[kbeg + i];
    __syncthreads();
    auto code_at = [&](int k) { return __builtin_amdgcn_readfirstlane(s_inc[k - kbeg]); };
#endif
    int buf = 0;
    A3R_STAMP(1);
    // one flat loop, two edge sides per trip, each register buffer re-requested right after it has been consumed (one edge side
    // in flight behind the one being worked on); the LDS batch of EB slots is flushed inside
    int kb = 0, k0 = kbeg;
#pragma unroll 1
    for (int k = kbeg; k < kend; k += A3R_ALIGN_NBUF) {
        const bool has1 = k + 1 < kend;
        consume(code_at(k), ea, buf, kb);
#ifdef A3R_ALIGN_STAMPS
        if (k == kbeg) { asm volatile("" :: "v"(gp[0][0])); A3R_STAMP(2); }
#endif
        if (k + A3R_ALIGN_NBUF < kend) load_edge(d, code_at(k + A3R_ALIGN_NBUF), P, pix0, valid, ea);
        if (has1) {
            consume(code_at(k + 1), eb, buf, kb + 1);
            if (k + 1 + A3R_ALIGN_NBUF < kend) load_edge(d, code_at(k + 1 + A3R_ALIGN_NBUF), P, pix0, valid, eb);
        }
#if A3R_ALIGN_NBUF == 3
        if (k + 2 < kend) {
            consume(code_at(k + 2), ec, buf, kb + 2);
            if (k + 5 < kend) load_edge(d, code_at(k + 5), P, pix0, valid, ec);
        }
#endif
        kb += A3R_ALIGN_NBUF;
        if (kb == EB || k + A3R_ALIGN_NBUF >= kend) {
            __syncthreads();
            if (tid < EB * 4) {
                // one 16-byte quarter of a slot's row per thread, rows r added in order (the row is handed to the image's last workgroup)
                const int sb = tid >> 2, q = tid & 3, ks = k0 + sb;
                if (ks < kend) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r++) s4 += *reinterpret_cast<const f32x4*>(&red[buf][sb][r][4 * q]);
                    store16_wt(d.partE, (unsigned)(((size_t)ks * d.nchunks + chunk) * 64 + 16 * q), s4);
                }
            }
            buf ^= 1; kb = 0; k0 += EB;
        }
    }
    A3R_STAMP(3);
    if (MODE != 0) {
    // the Adam moments of this thread's pixels are requested now: their latency runs under the per-image sums below
    f32x4 m4 = {0.f, 0.f, 0.f, 0.f}, v4 = {0.f, 0.f, 0.f, 0.f};
    if (VEC && MODE == 2 && valid[0]) {
        const size_t off = (size_t)n * P + pix0;
        m4 = *reinterpret_cast<const f32x4*>(d.adam_depth + off);
        v4 = *reinterpret_cast<const f32x4*>(d.adam_depth + (size_t)d.N * P + off);
    }

    // per-image sums and the per-pixel parameter (forward quantities are recomputed: cheaper than keeping them live)
    float accN[16], gout[PXT];
#pragma unroll
    for (int j = 0; j < 16; j++) accN[j] = 0.f;
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        float dep, ddp, gxm, gym, rel[3];
        pixel_forward(i, raw[i], monov[i], dep, ddp, gxm, gym, rel);
        const float h0 = R[0] * gp[i][0] + R[3] * gp[i][1] + R[6] * gp[i][2];
        const float h1 = R[1] * gp[i][0] + R[4] * gp[i][1] + R[7] * gp[i][2];
        const float h2 = R[2] * gp[i][0] + R[5] * gp[i][1] + R[8] * gp[i][2];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            accN[r * 3 + 0] += gp[i][r] * rel[0];
            accN[r * 3 + 1] += gp[i][r] * rel[1];
            accN[r * 3 + 2] += gp[i][r] * rel[2];
            accN[9 + r] += gp[i][r];
        }
        const float gd = h0 * gxm * inv_f + h1 * gym * inv_f + h2;
        accN[12] += -(h0 * rel[0] + h1 * rel[1]) / d.focal_break;
        accN[13] += -h0 * dep * inv_f * 10.f;
        accN[14] += -h1 * dep * inv_f * 10.f;
        accN[15] += gd;
        gout[i] = gd * ddp;
    }
    if (MODE != 0 && d.gprior) {      // depth prior (align_depth_prior_kernel): already w.r.t. the log-depth parameter
#pragma unroll
        for (int i = 0; i < PXT; i++)
            if (valid[i]) gout[i] += d.gprior[(size_t)n * P + pix0 + i * PSTEP];
    }
    const size_t NP = (size_t)d.N * P;
    if (VEC) {
        if (valid[0]) {
            const size_t off = (size_t)n * P + pix0;
            if (MODE == 1) {
                f32x4 g4 = {gout[0], gout[1], gout[2], gout[3]};
                *reinterpret_cast<f32x4*>(g_depth + off) = g4;
            } else {
                float pm[4] = {m4.x, m4.y, m4.z, m4.w}, pv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4];
#pragma unroll
                for (int i = 0; i < PXT; i++) { pp[i] = raw[i]; adam_update(pp[i], gout[i], pm[i], pv[i], ad); }
                f32x4 o0 = {pp[0], pp[1], pp[2], pp[3]}, o1 = {pm[0], pm[1], pm[2], pm[3]}, o2 = {pv[0], pv[1], pv[2], pv[3]};
                *reinterpret_cast<f32x4*>(d.depth + off) = o0;
                *reinterpret_cast<f32x4*>(d.adam_depth + off) = o1;
                *reinterpret_cast<f32x4*>(d.adam_depth + NP + off) = o2;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            if (!valid[i]) continue;
            const size_t off = (size_t)n * P + pix0 + i * PSTEP;
            if (MODE == 1) {
                g_depth[off] = gout[i];
            } else {
                float m = d.adam_depth[off], v = d.adam_depth[NP + off], pv = raw[i];
                adam_update(pv, gout[i], m, v, ad);
                d.depth[off] = pv; d.adam_depth[off] = m; d.adam_depth[NP + off] = v;
            }
        }
    }
    A3R_STAMP(4);
    __syncthreads();   // red[] may still be read by the last batch
    {
        float u[4];
        row_reduce_scatter16<16>(accN, u);
        if ((lane & 3) == 0) *reinterpret_cast<f32x4*>(&red[0][0][wave * 4 + (lane >> 4)][lane & 12]) = f32x4{u[0], u[1], u[2], u[3]};
    }
    __syncthreads();
    if (tid < 4) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r++) s4 += *reinterpret_cast<const f32x4*>(&red[0][0][r][4 * tid]);
        store16_wt(d.partN, (unsigned)(((size_t)n * d.nchunks + chunk) * 64 + 16 * tid), s4);
    }
    }   // MODE != 0
#ifdef A3R_ALIGN_STAMPS
    A3R_STAMP(5);
    if (tid == 0 && MODE == 2) {
        unsigned long long* o = g_align_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) % 8192 * 8;
        for (int i = 0; i < 6; i++) o[i] = stamp[i];
        o[6] = (unsigned long long)(kend - kbeg);
        o[7] = ((unsigned long long)n << 32) | (unsigned)chunk;
    }
#endif
    if (!d.fused_tail) return;

    // ---- tail of the iteration inside this launch (no finalize launches): last-block-done tickets, two levels.
    // Level 1, per image: the workgroup that completes image n adds the chunk partials of the image's incidence slots and of the
    // image itself in a fixed order (one wave per row set: the order does not depend on who runs it -> bitwise reproducible).
    __syncthreads();                                   // red[] is free again
    int* flag = reinterpret_cast<int*>(&red[0][0][0][0]);
    if (!arrive_last(d.tick + n, gridDim.x, flag)) return;
    for (int k = kbeg + wave; k < kend; k += TPB / 64) {
        const f32x4 t = wave_sum_rows(d.partE + (size_t)k * d.nchunks * 16, d.nchunks, lane);
        if (lane < 4) store16_wt(d.sumE, (unsigned)(k * 64 + 16 * lane), t);
    }
    if (MODE != 0 && wave == 0) {
        const f32x4 t = wave_sum_rows(d.partN + (size_t)n * d.nchunks * 16, d.nchunks, lane);
        if (lane < 4) store16_wt(d.sumN, (unsigned)(n * 64 + 16 * lane), t);
    }
    // Level 2: the workgroup that completes the last image runs the chain rules of all edges and images, then the single-block
    // finalisation (scale coupling, loss, Adam on the small parameters, next iteration's transforms).
    if (!arrive_last(d.tick + d.N, d.N, flag)) return;
    for (int e = tid; e < d.E; e += TPB) {
        const float* s0 = d.sumE + d.slot_of[e * 2 + 0] * 16;
        const float* s1 = d.sumE + d.slot_of[e * 2 + 1] * 16;
        double s[13];
#pragma unroll
        for (int j = 0; j < 13; j++) s[j] = (double)s0[j] + (double)s1[j];
        edge_chain(d, e, s, MODE == 0);
    }
    for (int m = tid; m < d.N; m += TPB) {
        double s[16];
#pragma unroll
        for (int j = 0; j < 16; j++) s[j] = (double)d.sumN[m * 16 + j];
        image_chain(d, m, s);
    }
    for (int i = tid; i <= d.N; i += TPB)              // counters back to zero for the next launch (write-through stores)
        __hip_atomic_store(d.tick + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                   // the chain-rule results are visible to the whole workgroup
    float* sh = &red[0][0][0][0];
    double (*shd)[4] = reinterpret_cast<double (*)[4]>(&red[1][0][0][0]);
    finalize_b_body<MODE>(d, ad, tout, sh, shd);
}

// ------------------------------------------------------------------------------------------- depth prior (cloud_opt_flow)
// depth_regularization_si_weighted (goem_opt.py:15-36) as optimizer.py:546-555 calls it: per image, with l = log clamp(depth, 1e-6),
// l0 the same of the initial depth map and w = 1 + dynamic_mask,
//     scale = sum(l0 - l) / HW,    loss_n = sum w (l - l0 + scale)^2 / HW,    prior = mean_n loss_n.
// One workgroup per image: pass 1 the four sums (fp64, fixed order), pass 2 the gradient w.r.t. the log-depth parameter, the
// dependence through `scale` included:  d loss_n / d l_q = (2 / HW) (w_q r_q - sum_p w_p r_p / HW).
constexpr int PRIOR_TPB = 1024;
constexpr float PRIOR_EPS = 1e-6f;
__global__ __launch_bounds__(PRIOR_TPB) void align_depth_prior_kernel(AlignDev d) {
    __shared__ double sh[4][PRIOR_TPB / 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int area = d.imarea[n];
    const float* raw = d.depth + (size_t)n * d.P;
    const float* raw0 = d.prior_init + (size_t)n * d.P;
    const unsigned char* dyn = d.prior_dyn ? d.prior_dyn + (size_t)n * d.P : nullptr;
    auto logd = [](float r) { return logf(fmaxf(expf(r), PRIOR_EPS)); };
    double acc[4] = {0.0, 0.0, 0.0, 0.0};        // sum delta, sum w delta, sum w delta^2, sum w
    for (int p = tid; p < area; p += PRIOR_TPB) {
        const float delta = logd(raw[p]) - logd(raw0[p]);
        const float w = (dyn && dyn[p]) ? 2.f : 1.f;
        acc[0] += delta; acc[1] += w * delta; acc[2] += (double)w * delta * delta; acc[3] += w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        if ((tid & 63) == 0) sh[k][tid >> 6] = acc[k];
    }
    __syncthreads();
    double tot[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        tot[k] = 0.0;
        for (int w = 0; w < PRIOR_TPB / 64; w++) tot[k] += sh[k][w];
    }
    const double inv = 1.0 / (double)area;
    const double scale = -tot[0] * inv;
    const double swr = tot[1] + scale * tot[3];                    // sum w (delta + scale)
    if (tid == 0) d.lossP[n] = (float)((tot[2] + 2.0 * scale * tot[1] + scale * scale * tot[3]) * inv);
    const float coef = d.prior_w / (float)d.N * 2.f * (float)inv, fscale = (float)scale, mean_wr = (float)(swr * inv);
    float* g = d.gprior + (size_t)n * d.P;
    for (int p = tid; p < d.P; p += PRIOR_TPB) {
        float v = 0.f;
        if (p < area && expf(raw[p]) > PRIOR_EPS) {                // the clamp passes no gradient below eps
            const float delta = logd(raw[p]) - logd(raw0[p]);
            const float w = (dyn && dyn[p]) ? 2.f : 1.f;
            v = coef * (w * (delta + fscale) - mean_wr);
        }
        g[p] = v;
    }
}

// ------------------------------------------------------------------------------------------- ego-flow term (cloud_opt_flow)
// optimizer.py:521-541 with DepthBasedWarping (goem_opt.py:195-236) and smooth_L1_loss_fn (optimizer.py:18-24).
// Image-major like the main kernel: the block's image is the SOURCE of every incident (edge, side); side 0 compares
// the ego-flow ei -> ej with flow_ij, side 1 the ego-flow ej -> ei with flow_ji.  Everything is UNSCALED by
// weight / sum(mask) (known only after the whole pass): per pixel the gradient w.r.t. the world point, per
// (edge, side) the loss sums and the gradient sums of the TARGET camera.
constexpr int NF = 17, NFP = 20;    // per-slot sums: S, C, d/df_t, d/dcx_t, d/dcy_t, sum gY[3], sum v (x) gY [9]

__global__ __launch_bounds__(TPB) void align_flow_kernel(AlignDev d, const int* __restrict__ inc_ptr, const int* __restrict__ inc,
                                                          const int* __restrict__ other, const float* __restrict__ img_xf) {
    __shared__ float red[2][EB][16][NFP];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = d.P, W = d.fW;
    const float* ix = img_xf + n * 16;
    float Rs[9], Ts[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { Rs[r * 3] = ix[r * 4]; Rs[r * 3 + 1] = ix[r * 4 + 1]; Rs[r * 3 + 2] = ix[r * 4 + 2]; Ts[r] = ix[r * 4 + 3]; }
    const float fs = ix[12], cxs = ix[13], cys = ix[14];
    float px[PXT], py[PXT], dpv[PXT], Pw[PXT][3], g0[PXT][3], g1[PXT][3];
    bool ok[PXT];
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const int p = chunk * CHUNK + i * TPB + tid;
        const bool in = p < P;
        ok[i] = in && d.dyn[(size_t)n * P + p] == 0;          // mask = ~dynamic_mask of the source image
        const float dp = (in ? expf(d.depth[(size_t)n * P + p]) : 1.f) + 1e-6f;   // 1 / disp
        dpv[i] = dp;
        const int y = p / W, x = p - y * W;
        px[i] = (float)x; py[i] = (float)y;
        const float X0 = dp * (px[i] - cxs) / fs, X1 = dp * (py[i] - cys) / fs, X2 = dp;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Pw[i][r] = Rs[r * 3] * X0 + Rs[r * 3 + 1] * X1 + Rs[r * 3 + 2] * X2 + Ts[r];
            g0[i][r] = 0.f; g1[i][r] = 0.f;
        }
    }
    const int kbeg = inc_ptr[n], kend = inc_ptr[n + 1];
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += EB) {
#pragma unroll 1
        for (int kb = 0; kb < EB; kb++) {
            const int k = k0 + kb;
            if (k >= kend) break;
            const int code = inc[k], e = code >> 1, side = code & 1, t = other[k];
            const float* tx = img_xf + t * 16;
            const float r00 = tx[0], r01 = tx[1], r02 = tx[2], t0 = tx[3];
            const float r10 = tx[4], r11 = tx[5], r12 = tx[6], t1 = tx[7];
            const float r20 = tx[8], r21 = tx[9], r22 = tx[10], t2 = tx[11];
            const float ft = tx[12], cxt = tx[13], cyt = tx[14];
            const float* fl = (side ? d.flow_ji : d.flow_ij) + (size_t)e * 2 * P;
            float acc[NF];
#pragma unroll
            for (int j = 0; j < NF; j++) acc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < PXT; i++) {
                const int p = chunk * CHUNK + i * TPB + tid;
                const float gt0 = ok[i] ? fl[p] : 0.f, gt1 = ok[i] ? fl[P + p] : 0.f;
                const float v0 = Pw[i][0] - t0, v1 = Pw[i][1] - t1, v2 = Pw[i][2] - t2;
                const float Y0 = r00 * v0 + r10 * v1 + r20 * v2;        // Y = R_t^T v
                const float Y1 = r01 * v0 + r11 * v1 + r21 * v2;
                const float Y2 = r02 * v0 + r12 * v1 + r22 * v2;
                const float qx = ft * Y0 + cxt * Y2, qy = ft * Y1 + cyt * Y2;
                // the reference normalises disp*K*Y by (disp*z + 1e-6), i.e. K*Y by (z + 1e-6 / disp)
                const float den = Y2 + 1e-6f * dpv[i];
                const float iz = 1.f / den;
                const float e0 = qx * iz - px[i], e1 = qy * iz - py[i];
                float gn0 = 0.f, gn1 = 0.f;
                if (ok[i]) {
                    const float d0 = e0 - gt0, a0 = fabsf(d0), l0 = a0 < 1.f ? 0.5f * d0 * d0 : a0 - 0.5f;
                    const float d1 = e1 - gt1, a1 = fabsf(d1), l1 = a1 < 1.f ? 0.5f * d1 * d1 : a1 - 0.5f;
                    if (l0 < d.pxl_thre) { acc[0] += l0; acc[1] += 1.f; gn0 = a0 < 1.f ? d0 : (d0 > 0.f ? 1.f : -1.f); }
                    if (l1 < d.pxl_thre) { acc[0] += l1; acc[1] += 1.f; gn1 = a1 < 1.f ? d1 : (d1 > 0.f ? 1.f : -1.f); }
                }
                const float gq0 = gn0 * iz, gq1 = gn1 * iz, gq2 = -(gn0 * qx + gn1 * qy) * iz * iz;
                const float gY0 = ft * gq0, gY1 = ft * gq1, gY2 = cxt * gq0 + cyt * gq1 + gq2;
                acc[2] += gq0 * Y0 + gq1 * Y1;
                acc[3] += gq0 * Y2;
                acc[4] += gq1 * Y2;
                acc[5] += gY0; acc[6] += gY1; acc[7] += gY2;
                acc[8] += v0 * gY0; acc[9] += v0 * gY1; acc[10] += v0 * gY2;
                acc[11] += v1 * gY0; acc[12] += v1 * gY1; acc[13] += v1 * gY2;
                acc[14] += v2 * gY0; acc[15] += v2 * gY1; acc[16] += v2 * gY2;
                const float gw0 = r00 * gY0 + r01 * gY1 + r02 * gY2;     // gPw = R_t gY
                const float gw1 = r10 * gY0 + r11 * gY1 + r12 * gY2;
                const float gw2 = r20 * gY0 + r21 * gY1 + r22 * gY2;
                if (side) { g1[i][0] += gw0; g1[i][1] += gw1; g1[i][2] += gw2; }
                else { g0[i][0] += gw0; g0[i][1] += gw1; g0[i][2] += gw2; }
            }
#pragma unroll
            for (int j = 0; j < NF; j++) {
                const float s = dpp_row_sum16(acc[j]);
                if ((lane & 15) == 0) red[buf][kb][wave * 4 + (lane >> 4)][j] = s;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < EB * NFP; idx += TPB) {
            const int kb = idx / NFP, j = idx - kb * NFP, k = k0 + kb;
            if (k < kend && j < NF) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) s += red[buf][kb][r][j];
                d.partF[((size_t)k * d.nchunks + chunk) * NFP + j] = s;
            }
        }
        buf ^= 1;
    }
    const size_t NP3 = (size_t)d.N * P * 3;
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const int p = chunk * CHUNK + i * TPB + tid;
        if (p >= P) continue;
        float* o = d.gflow + ((size_t)n * P + p) * 3;
#pragma unroll
        for (int r = 0; r < 3; r++) { o[r] = g0[i][r]; o[NP3 + r] = g1[i][r]; }
    }
}

// The same pass for P % 4 == 0 (every model resolution), restructured like the main kernel (round 3): a thread owns FOUR CONSECUTIVE
// pixels, so an edge side's flow is two 16-byte loads per thread, requested one edge side ahead into a second register pair; the
// incidence codes, the partner image and its transform come by scalar loads; the 17 per-slot sums use the DPP reduce-scatter
// (16 of them: 32 VALU + one 16-byte LDS store per quad) instead of 17 four-step butterflies with masked stores.  The first form
// was bound by its own instruction stream (2.2 TB/s on BASELINE config 4's problem: 2.05 ms of a 3.96 ms iteration).
__global__ __launch_bounds__(TPB, 4) void align_flow_vec_kernel(AlignDev d, const int* __restrict__ inc_ptr, const int* __restrict__ inc,
                                                              const int* __restrict__ other, const float* __restrict__ img_xf) {
    __shared__ float red[2][EB][16][NFP];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = d.P, W = d.fW;
    const int p0 = chunk * CHUNK + tid * PXT;
    const bool in = p0 < P;                                   // the four pixels are in or out together (P % 4 == 0)
    const int kbeg = inc_ptr[n], kend = inc_ptr[n + 1];
    auto load_flow = [&](int k, f32x4& fx, f32x4& fy) {
        const int code = inc[__builtin_amdgcn_readfirstlane(k)];
        const float* fl = ((code & 1) ? d.flow_ji : d.flow_ij) + (size_t)(code >> 1) * 2 * P + (in ? p0 : 0);
        fx = *reinterpret_cast<const f32x4*>(fl);
        fy = *reinterpret_cast<const f32x4*>(fl + P);
    };
    f32x4 fxa, fya, fxb, fyb;
    f32x4 dep4 = {0.f, 0.f, 0.f, 0.f};
    unsigned dyn4 = 0xffffffffu;
    if (in) {
        dep4 = *reinterpret_cast<const f32x4*>(d.depth + (size_t)n * P + p0);
        dyn4 = *reinterpret_cast<const unsigned*>(d.dyn + (size_t)n * P + p0);
    }
    if (kbeg < kend) load_flow(kbeg, fxa, fya);               // right behind the depth: the set-up arithmetic runs under its latency
    const float* ix = img_xf + n * 16;
    float Rs[9], Ts[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { Rs[r * 3] = ix[r * 4]; Rs[r * 3 + 1] = ix[r * 4 + 1]; Rs[r * 3 + 2] = ix[r * 4 + 2]; Ts[r] = ix[r * 4 + 3]; }
    const float fs = ix[12], cxs = ix[13], cys = ix[14];
    float px[PXT], py[PXT], dpv[PXT], Pw[PXT][3], g0[PXT][3], g1[PXT][3];
    bool ok[PXT];
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const int p = p0 + i;
        ok[i] = in && ((dyn4 >> (8 * i)) & 0xffu) == 0;        // mask = ~dynamic_mask of the source image
        const float dp = (in ? expf(dep4[i]) : 1.f) + 1e-6f;   // 1 / disp
        dpv[i] = dp;
        const int y = p / W, x = p - y * W;
        px[i] = (float)x; py[i] = (float)y;
        const float X0 = dp * (px[i] - cxs) / fs, X1 = dp * (py[i] - cys) / fs, X2 = dp;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Pw[i][r] = Rs[r * 3] * X0 + Rs[r * 3 + 1] * X1 + Rs[r * 3 + 2] * X2 + Ts[r];
            g0[i][r] = 0.f; g1[i][r] = 0.f;
        }
    }
    auto consume = [&](int k, const f32x4& fx, const f32x4& fy, int buf, int kb) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
        const int side = inc[ku] & 1, t = other[ku];
        const float* tx = img_xf + t * 16;
        const float r00 = tx[0], r01 = tx[1], r02 = tx[2], t0 = tx[3];
        const float r10 = tx[4], r11 = tx[5], r12 = tx[6], t1 = tx[7];
        const float r20 = tx[8], r21 = tx[9], r22 = tx[10], t2 = tx[11];
        const float ft = tx[12], cxt = tx[13], cyt = tx[14];
        float acc[NF], gw[PXT][3];
#pragma unroll
        for (int j = 0; j < NF; j++) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const float gt0 = ok[i] ? fx[i] : 0.f, gt1 = ok[i] ? fy[i] : 0.f;
            const float v0 = Pw[i][0] - t0, v1 = Pw[i][1] - t1, v2 = Pw[i][2] - t2;
            const float Y0 = r00 * v0 + r10 * v1 + r20 * v2;        // Y = R_t^T v
            const float Y1 = r01 * v0 + r11 * v1 + r21 * v2;
            const float Y2 = r02 * v0 + r12 * v1 + r22 * v2;
            const float qx = ft * Y0 + cxt * Y2, qy = ft * Y1 + cyt * Y2;
            const float den = Y2 + 1e-6f * dpv[i];
#ifdef A3R_FLOW_IEEE_DIV
            const float iz = 1.f / den;
#else
            // v_rcp_f32 (1 ulp) + one Newton step instead of the IEEE division's expansion (a dozen instructions per pixel side)
            const float r0 = __builtin_amdgcn_rcpf(den);
            const float iz = r0 * (2.f - den * r0);
#endif
            const float e0 = qx * iz - px[i], e1 = qy * iz - py[i];
            float gn0 = 0.f, gn1 = 0.f;
            if (ok[i]) {
                const float d0 = e0 - gt0, a0 = fabsf(d0), l0 = a0 < 1.f ? 0.5f * d0 * d0 : a0 - 0.5f;
                const float d1 = e1 - gt1, a1 = fabsf(d1), l1 = a1 < 1.f ? 0.5f * d1 * d1 : a1 - 0.5f;
                if (l0 < d.pxl_thre) { acc[0] += l0; acc[1] += 1.f; gn0 = a0 < 1.f ? d0 : (d0 > 0.f ? 1.f : -1.f); }
                if (l1 < d.pxl_thre) { acc[0] += l1; acc[1] += 1.f; gn1 = a1 < 1.f ? d1 : (d1 > 0.f ? 1.f : -1.f); }
            }
            const float gq0 = gn0 * iz, gq1 = gn1 * iz, gq2 = -(gn0 * qx + gn1 * qy) * iz * iz;
            const float gY0 = ft * gq0, gY1 = ft * gq1, gY2 = cxt * gq0 + cyt * gq1 + gq2;
            acc[2] += gq0 * Y0 + gq1 * Y1;
            acc[3] += gq0 * Y2;
            acc[4] += gq1 * Y2;
            acc[5] += gY0; acc[6] += gY1; acc[7] += gY2;
            acc[8] += v0 * gY0; acc[9] += v0 * gY1; acc[10] += v0 * gY2;
            acc[11] += v1 * gY0; acc[12] += v1 * gY1; acc[13] += v1 * gY2;
            acc[14] += v2 * gY0; acc[15] += v2 * gY1; acc[16] += v2 * gY2;
            gw[i][0] = r00 * gY0 + r01 * gY1 + r02 * gY2;            // gPw = R_t gY
            gw[i][1] = r10 * gY0 + r11 * gY1 + r12 * gY2;
            gw[i][2] = r20 * gY0 + r21 * gY1 + r22 * gY2;
        }
        if (side) {                                                  // wave-uniform (a scalar load): a scalar branch, no selects
#pragma unroll
            for (int i = 0; i < PXT; i++) { g1[i][0] += gw[i][0]; g1[i][1] += gw[i][1]; g1[i][2] += gw[i][2]; }
        } else {
#pragma unroll
            for (int i = 0; i < PXT; i++) { g0[i][0] += gw[i][0]; g0[i][1] += gw[i][1]; g0[i][2] += gw[i][2]; }
        }
        float v16[16], u[4];
#pragma unroll
        for (int j = 0; j < 16; j++) v16[j] = acc[j];
        row_reduce_scatter16<16>(v16, u);
        const float s16 = dpp_row_sum16(acc[16]);
        float* row = &red[buf][kb][wave * 4 + (lane >> 4)][0];
        if ((lane & 3) == 0) *reinterpret_cast<f32x4*>(row + (lane & 12)) = f32x4{u[0], u[1], u[2], u[3]};
        if ((lane & 15) == 0) row[16] = s16;
    };
    int buf = 0, kb = 0, k0 = kbeg;
#pragma unroll 1
    for (int k = kbeg; k < kend; k += 2) {
        const bool has1 = k + 1 < kend;
        if (has1) load_flow(k + 1, fxb, fyb);
        consume(k, fxa, fya, buf, kb);
        if (has1) {
            if (k + 2 < kend) load_flow(k + 2, fxa, fya);
            consume(k + 1, fxb, fyb, buf, kb + 1);
        }
        kb += 2;
        if (kb == EB || k + 2 >= kend) {
            __syncthreads();
            for (int idx = tid; idx < EB * NFP; idx += TPB) {
                const int sb = idx / NFP, j = idx - sb * NFP, ks = k0 + sb;
                if (ks < kend && j < NF) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r++) s += red[buf][sb][r][j];
                    d.partF[((size_t)ks * d.nchunks + chunk) * NFP + j] = s;
                }
            }
            buf ^= 1; kb = 0; k0 += EB;
        }
    }
    if (in) {
        const size_t NP3 = (size_t)d.N * P * 3;
        f32x4* o = reinterpret_cast<f32x4*>(d.gflow + ((size_t)n * P + p0) * 3);
        f32x4* o1 = reinterpret_cast<f32x4*>(d.gflow + NP3 + ((size_t)n * P + p0) * 3);
        o[0] = f32x4{g0[0][0], g0[0][1], g0[0][2], g0[1][0]};
        o[1] = f32x4{g0[1][1], g0[1][2], g0[2][0], g0[2][1]};
        o[2] = f32x4{g0[2][2], g0[3][0], g0[3][1], g0[3][2]};
        o1[0] = f32x4{g1[0][0], g1[0][1], g1[0][2], g1[1][0]};
        o1[1] = f32x4{g1[1][1], g1[1][2], g1[2][0], g1[2][1]};
        o1[2] = f32x4{g1[2][2], g1[3][0], g1[3][1], g1[3][2]};
    }
}

// grid 2E x 64 threads: fixed-order sum of the chunk partials of one incidence slot
__global__ __launch_bounds__(64) void align_flow_reduce_kernel(AlignDev d) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (lane >= NFP) return;
    float s = 0.f;
    for (int c = 0; c < d.nchunks; c++) s += d.partF[((size_t)k * d.nchunks + c) * NFP + lane];
    d.sumF[k * NFP + lane] = lane < NF ? s : 0.f;
}

// one block: the two normalisers, the loss value, the drop decision (optimizer.py:536-540)
__global__ __launch_bounds__(TPB) void align_flow_decide_kernel(AlignDev d, const int* __restrict__ inc) {
    __shared__ double sh[4][4];
    const int tid = threadIdx.x;
    double acc[4] = {0, 0, 0, 0};
    for (int k = tid; k < 2 * d.E; k += TPB) {
        const int dir = inc[k] & 1;
        acc[dir * 2] += (double)d.sumF[k * NFP];
        acc[dir * 2 + 1] += (double)d.sumF[k * NFP + 1];
    }
    for (int j = 0; j < 4; j++) {
        double v = acc[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) sh[j][tid >> 6] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double t[4];
        for (int j = 0; j < 4; j++) t[j] = sh[j][0] + sh[j][1] + sh[j][2] + sh[j][3];
        const float loss = (float)(t[0] / t[1] + t[2] / t[3]);
        const bool dropped = loss > d.flow_thre && d.flow_thre > 0.f;
        d.flow_state[0] = dropped ? 0.f : (float)(d.flow_w / t[1]);
        d.flow_state[1] = dropped ? 0.f : (float)(d.flow_w / t[3]);
        d.flow_state[2] = loss;
        d.flow_state[3] = dropped ? 1.f : 0.f;
        if (dropped) d.flow_state[4] = 1.f;          // sticky, like self.flow_loss_flag
    }
}

// relative_pose_loss between images a and b = a + 1 (optimizer.py:559-572): loss, and gradient w.r.t. rotation /
// translation of `which` (0: a, 1: b) accumulated into GR[9], GT[3].
__device__ double temporal_pair(const float* Ra, const float* Ta, const float* Rb, const float* Tb, float tw, int which,
                                double* GR, double* GT) {
    double M[9], a = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)Ra[k * 3 + i] * Rb[k * 3 + j];
            M[i * 3 + j] = s - (i == j);
            a += M[i * 3 + j] * M[i * 3 + j];
        }
    a = sqrt(a);
    double dv[3], u[3], un = 0;
    for (int k = 0; k < 3; k++) dv[k] = (double)Tb[k] - Ta[k];
    for (int i = 0; i < 3; i++) { u[i] = Ra[i] * dv[0] + Ra[3 + i] * dv[1] + Ra[6 + i] * dv[2]; un += u[i] * u[i]; }
    un = sqrt(un);
    if (a > 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += which ? (double)Ra[i * 3 + k] * M[k * 3 + j] / a : (double)Rb[i * 3 + k] * M[j * 3 + k] / a;
                GR[i * 3 + j] += s;
            }
    if (un > 0) {
        double gu[3], Rg[3];
        for (int i = 0; i < 3; i++) gu[i] = tw * u[i] / un;
        for (int i = 0; i < 3; i++) Rg[i] = Ra[i * 3] * gu[0] + Ra[i * 3 + 1] * gu[1] + Ra[i * 3 + 2] * gu[2];
        for (int i = 0; i < 3; i++) {
            GT[i] += which ? Rg[i] : -Rg[i];
            if (!which) for (int j = 0; j < 3; j++) GR[i * 3 + j] += dv[i] * gu[j];
        }
    }
    return a + tw * un;
}

// ------------------------------------------------------------------------------------------- finalize A
// Chain rule of one edge from its 13 sums (both sides added): loss, d/d pw_pose (quaternion, translation, log-scale before the
// mean coupling) and d/d pw_adaptors.
__device__ void edge_chain(const AlignDev& d, int e, const double* s, bool loss_only) {
    d.lossE[e] = (float)s[12];
    if (loss_only) return;
    const float* p = d.pw_poses + e * 8;
    const float* xf = d.edge_xf + e * 16;
    float R[9], qn[4], nrm;
    quat_to_R(p, R, qn, &nrm);
    const double sc = xf[12];
    const float a[3] = {xf[13], xf[14], xf[15]};
    double G[9], dLds = 0.0;
    for (int r = 0; r < 3; r++) {
        for (int q = 0; q < 3; q++) {
            G[r * 3 + q] = -sc * a[q] * s[r * 3 + q];
            dLds -= (double)R[r * 3 + q] * a[q] * s[r * 3 + q];
        }
        dLds -= (double)signed_expm1f(p[4 + r]) * s[9 + r];
    }
    double gq[4];
    quat_backward(qn, nrm, G, gq);
    float* g = d.gE + e * 8;
    for (int k = 0; k < 4; k++) g[k] = (float)gq[k];
    for (int k = 0; k < 3; k++) g[4 + k] = (float)(-sc * s[9 + k] * signed_expm1_grad(p[4 + k]));
    g[7] = (float)(dLds * sc);   // S_e * s_e; the mean coupling is applied in finalize B
    // pw_adaptors (base_opt.py:177-182): aligned = s R diag(a) X + s T with a = exp(adapt / pw_break),
    // adapt = (p0, p0, p1) [- its mean when norm_pw_scale].  dL/da_c = - sum_r s R_rc (sum g_r X_c).
    double ga[3], gmean = 0.0;
    for (int c = 0; c < 3; c++) {
        double t = 0.0;
        for (int r = 0; r < 3; r++) t -= sc * (double)R[r * 3 + c] * s[r * 3 + c];
        ga[c] = t * a[c] / d.pw_break;                   // w.r.t. the (centred) exponent
        gmean += ga[c];
    }
    if (d.norm_pw_scale) { gmean /= 3.0; for (int c = 0; c < 3; c++) ga[c] -= gmean; }
    d.gA[e * 2 + 0] = (float)(ga[0] + ga[1]);
    d.gA[e * 2 + 1] = (float)ga[2];
}

// Chain rule of one image from its 16 sums (+ the ego-flow target-camera terms and the temporal smoothing term).
// (in loss-only launches the partial sums are stale: only lossN is consumed)
__device__ void image_chain(const AlignDev& d, int n, double* s) {
    const float* p = d.im_poses + n * 7;
    float R[9], qn[4], nrm;
    quat_to_R(p, R, qn, &nrm);
    if (d.flow_on) {
        // image n as the TARGET camera of the ego-flow of the opposite side of each incident edge
        const float fn = d.img_xf[n * 16 + 12];
        for (int k = d.inc_ptr[n]; k < d.inc_ptr[n + 1]; k++) {
            const int code = d.inc[k], opp = d.slot_of[(code >> 1) * 2 + (1 - (code & 1))];
            const double c = d.flow_state[1 - (code & 1)];
            const float* F = d.sumF + opp * NFP;
            for (int j = 0; j < 9; j++) s[j] += c * F[8 + j];                       // dL/dR_t = v (x) gY
            for (int i = 0; i < 3; i++)
                s[9 + i] -= c * ((double)R[i * 3] * F[5] + (double)R[i * 3 + 1] * F[6] + (double)R[i * 3 + 2] * F[7]);
            s[12] += c * F[2] * fn / d.focal_break;
            s[13] += 10.0 * c * F[3];
            s[14] += 10.0 * c * F[4];
        }
    }
    if (d.tsw > 0.f) {
        float Tn[3], Ro[9], To[3];
        for (int k = 0; k < 3; k++) Tn[k] = signed_expm1f(p[4 + k]);
        double GR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, GT[3] = {0, 0, 0};
        if (n + 1 < d.N) {
            const float* po = d.im_poses + (n + 1) * 7;
            quat_to_R(po, Ro, nullptr, nullptr);
            for (int k = 0; k < 3; k++) To[k] = signed_expm1f(po[4 + k]);
            d.lossN[n] = (float)(d.tsw * temporal_pair(R, Tn, Ro, To, d.trans_w, 0, GR, GT));
        } else {
            d.lossN[n] = 0.f;
        }
        if (n > 0) {
            const float* po = d.im_poses + (n - 1) * 7;
            quat_to_R(po, Ro, nullptr, nullptr);
            for (int k = 0; k < 3; k++) To[k] = signed_expm1f(po[4 + k]);
            temporal_pair(Ro, To, R, Tn, d.trans_w, 1, GR, GT);
        }
        for (int j = 0; j < 9; j++) s[j] += d.tsw * GR[j];
        for (int j = 0; j < 3; j++) s[9 + j] += d.tsw * GT[j];
    }
    double gq[4];
    quat_backward(qn, nrm, s, gq);
    float* g = d.gN + n * 16;
    for (int k = 0; k < 4; k++) g[k] = (float)gq[k];
    for (int k = 0; k < 3; k++) g[4 + k] = (float)(s[9 + k] * signed_expm1_grad(p[4 + k]));
    g[7] = (float)s[12]; g[8] = (float)s[13]; g[9] = (float)s[14]; g[10] = (float)s[15];
    for (int k = 11; k < 16; k++) g[k] = 0.f;
}

// One wave: fixed-order sum of the nchunks 16-float partial rows at `base`.  lane = 4 c' + q reads floats [4q, 4q+4) of chunks
// c', c' + 16, ... (16-byte loads, eight in flight: the rows come from memory, not from this XCD's L2), then the 16 lanes that
// share q are added by xor-shuffles (4, 8, 16, 32): the order never depends on which wave or workgroup runs it.
// Returns, in every lane, the totals of floats [4q, 4q+4) with q = lane & 3.
__device__ __forceinline__ f32x4 wave_sum_rows(const float* base, int nchunks, int lane) {
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    const int q = lane & 3;
    const f32x4* pe = reinterpret_cast<const f32x4*>(base);
    for (int c0 = lane >> 2; c0 < nchunks; c0 += 16 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = c0 + 16 * u;
            v[u] = pe[(c < nchunks ? c : c0) * 4 + q];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (c0 + 16 * u < nchunks) part += v[u];
    }
    float pv[4] = {part.x, part.y, part.z, part.w};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float v = pv[t];
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        pv[t] = v;
    }
    const f32x4 r = {pv[0], pv[1], pv[2], pv[3]};
    return r;
}
// the 16 totals in every lane (float j lives in element j & 3 of lane j >> 2)
__device__ __forceinline__ void quad_to_all(f32x4 t, float* tot /*[16]*/) {
    const float pv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 16; j++) tot[j] = __shfl(pv[j & 3], j >> 2);
}

// Separate-launch form (A3R_ALIGN_TAIL=launch): grid E + N blocks of 64 threads.
__global__ __launch_bounds__(64) void align_finalize_a_kernel(AlignDev d, int loss_only) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float t0[16], t1[16];
    if (b < d.E) {
        quad_to_all(wave_sum_rows(d.partE + (size_t)d.slot_of[b * 2 + 0] * d.nchunks * 16, d.nchunks, lane), t0);
        quad_to_all(wave_sum_rows(d.partE + (size_t)d.slot_of[b * 2 + 1] * d.nchunks * 16, d.nchunks, lane), t1);
        if (lane == 0) {
            double s[13];
            for (int j = 0; j < 13; j++) s[j] = (double)t0[j] + (double)t1[j];
            edge_chain(d, b, s, loss_only != 0);
        }
    } else {
        const int n = b - d.E;
        quad_to_all(wave_sum_rows(d.partN + (size_t)n * d.nchunks * 16, d.nchunks, lane), t0);
        if (lane == 0) {
            double s[16];
            for (int j = 0; j < 16; j++) s[j] = (double)t0[j];
            image_chain(d, n, s);
        }
    }
}

// ------------------------------------------------------------------------------------------- finalize B
// one block: scale-normalisation coupling, loss, Adam on the small parameters, transforms for the next iteration.
// MODE 0: loss only -> loss_out; 1: gradients -> g_pw / g_small / loss_out; 2: update.
template <int MODE>
__device__ void finalize_b_body(const AlignDev& d, const AdamArgs& ad, const TailOut& o, float* sh /*[TPB]*/, double (*shd)[4] /*[2][4]*/) {
    float* g_pw = o.g_pw; float* g_small = o.g_small; float* loss_out = o.loss_out; float* g_adapt = o.g_adapt;
    const int tid = threadIdx.x;
    double lsum = 0.0, ssum = 0.0;
    for (int e = tid; e < d.E; e += TPB) {
        lsum += (double)d.lossE[e];
        if (MODE != 0) ssum += (double)d.gE[e * 8 + 7];
    }
    for (int o = 32; o > 0; o >>= 1) { lsum += __shfl_xor(lsum, o); ssum += __shfl_xor(ssum, o); }
    if ((tid & 63) == 0) { shd[0][tid >> 6] = lsum; shd[1][tid >> 6] = ssum; }
    __syncthreads();
    double loss = shd[0][0] + shd[0][1] + shd[0][2] + shd[0][3];
    const double sumSs = shd[1][0] + shd[1][1] + shd[1][2] + shd[1][3];
    if (tid == 0) {
        if (d.tsw > 0.f) for (int n = 0; n + 1 < d.N; n++) loss += (double)d.lossN[n];
        if (d.flow_on && d.flow_state[3] == 0.f) loss += (double)d.flow_w * d.flow_state[2];
        if (d.prior_w > 0.f) {
            double lp = 0.0;
            for (int n = 0; n < d.N; n++) lp += (double)d.lossP[n];
            loss += (double)d.prior_w * lp / d.N;
        }
        if (MODE == 2) d.loss_history[ad.step] = (float)loss;
        else *loss_out = (float)loss;
    }
    if (MODE == 0) return;
    const float corr = d.norm_pw_scale ? (float)(sumSs / d.E) : 0.f;
    for (int i = tid; i < d.E * 8; i += TPB) {
        float g = d.gE[i];
        if ((i & 7) == 7) g -= corr;
        if (MODE == 1) {
            g_pw[i] = g;
        } else {
            float m = d.adam_pw_poses[i], v = d.adam_pw_poses[d.E * 8 + i], p = d.pw_poses[i];
            adam_update(p, g, m, v, ad);
            d.pw_poses[i] = p; d.adam_pw_poses[i] = m; d.adam_pw_poses[d.E * 8 + i] = v;
        }
    }
    for (int i = tid; i < d.E * 2; i += TPB) {
        const float g = d.gA[i];
        if (MODE == 1) {
            if (g_adapt) g_adapt[i] = g;
        } else if (d.train_adaptors) {
            float m = d.adam_pw_adaptors[i], v = d.adam_pw_adaptors[d.E * 2 + i], p = d.pw_adaptors[i];
            adam_update(p, g, m, v, ad);
            d.pw_adaptors[i] = p; d.adam_pw_adaptors[i] = m; d.adam_pw_adaptors[d.E * 2 + i] = v;
        }
    }
    for (int i = tid; i < d.N * 16; i += TPB) {
        const int n = i >> 4, j = i & 15;
        const float g = d.gN[i];
        if (MODE == 1) { g_small[i] = g; continue; }
        float* target = nullptr;
        if (j < 7) { if (d.train_poses) target = d.im_poses + n * 7 + j; }
        else if (j == 7) { if (d.train_focals && !d.shared_focal) target = d.im_focals + n; }
        else if (j < 10) { if (d.train_pp) target = d.im_pp + n * 2 + (j - 8); }
        else if (j == 10) { if (d.mono) target = d.shifts + n; }
        if (target) {
            float m = d.adam_small[i], v = d.adam_small[d.N * 16 + i], p = *target;
            adam_update(p, g, m, v, ad);
            *target = p; d.adam_small[i] = m; d.adam_small[d.N * 16 + i] = v;
        }
    }
    if (MODE == 2 && d.shared_focal && d.train_focals && tid == 0) {
        // one focal parameter shared by every image (optimizer.py:56-58): its gradient is the sum over images
        double g = 0.0;
        for (int n = 0; n < d.N; n++) g += (double)d.gN[n * 16 + 7];
        float m = d.adam_small[7], v = d.adam_small[d.N * 16 + 7], p = d.im_focals[0];
        adam_update(p, (float)g, m, v, ad);
        d.im_focals[0] = p; d.adam_small[7] = m; d.adam_small[d.N * 16 + 7] = v;
    }
    if (MODE == 2) {
        __syncthreads();          // parameter stores above are visible to the whole workgroup
        build_transforms(d, sh);
    }
}

template <int MODE>
__global__ __launch_bounds__(TPB) void align_finalize_b_kernel(AlignDev d, AdamArgs ad, TailOut o) {
    __shared__ float sh[TPB];
    __shared__ double shd[2][4];
    finalize_b_body<MODE>(d, ad, o, sh, shd);
}

__global__ void align_export_xf_kernel(AlignDev d, float* edge_M, float* img_R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.E * 12) edge_M[i] = d.edge_xf[(i / 12) * 16 + i % 12];
    if (i < d.N * 12) img_R[i] = d.img_xf[(i / 12) * 16 + i % 12];
}

}  // namespace a3r

// =============================================================================================== host
using namespace a3r;

struct a3r_align_s {
    AlignDev d;
    bool use_mono, dist_l2;
    int steps;
    int loss_capacity;
    bool dirty;      // parameters changed by the caller since the transforms were last built
    std::vector<int> ei, ej, inc;     // host copies of the graph (for a3r_align_set_flow)
    int flow_start_iter = 0;
};

static void refresh_if_dirty(a3r_align_s* a, hipStream_t st) {
    if (a->dirty) {
        hipLaunchKernelGGL(align_prep_kernel, dim3(1), dim3(TPB), 0, st, a->d);
        a->dirty = false;
    }
}

static size_t ws_layout(int E, int N, int P, size_t* off /*[17]*/) {
    const int nch = (P + CHUNK - 1) / CHUNK;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    off[0] = take((size_t)E * 16 * 4);              // edge_xf
    off[1] = take((size_t)N * 16 * 4);              // img_xf
    off[2] = take((size_t)2 * E * nch * 16 * 4);    // partE
    off[3] = take((size_t)N * nch * 16 * 4);        // partN
    off[4] = take((size_t)E * 8 * 4);               // gE
    off[5] = take((size_t)N * 16 * 4);              // gN
    off[6] = take((size_t)E * 4);                   // lossE
    off[7] = take((size_t)(N + 1) * 4);             // inc_ptr
    off[8] = take((size_t)2 * E * 4);               // inc
    off[9] = take((size_t)2 * E * 4);               // slot_of
    off[10] = take((size_t)N * 4);                  // imw
    off[11] = take((size_t)N * 4);                  // imarea
    off[12] = take((size_t)E * 2 * 4);              // gA
    off[13] = take((size_t)2 * E * 16 * 4);         // sumE
    off[14] = take((size_t)N * 16 * 4);             // sumN
    off[15] = take((size_t)(N + 1) * 4);            // tick
    off[16] = take((size_t)N * 8 * 4);              // order: per dispatch slot {image, kbeg, kend, code0, code1, code2, 0, 0}
    return o;
}

extern "C" size_t a3r_align_workspace_bytes(int E, int N, int P) {
    size_t off[17];
    return ws_layout(E, N, P, off);
}

extern "C" int a3r_align_create(const a3r_align_desc* s, a3r_align_t* out, void* stream) {
    A3R_CHECK_ARG(s && out, "a3r_align_create: null argument");
    A3R_CHECK_ARG(s->E > 0 && s->N > 0 && s->P > 0, "a3r_align_create: E, N, P must be positive");
    A3R_CHECK_ARG(s->pred_i && s->pred_j && s->w_i && s->w_j && s->pp0, "a3r_align_create: missing observation buffers");
    A3R_CHECK_ARG(s->pw_poses && s->pw_adaptors && s->depth && s->im_poses && s->im_focals && s->im_pp,
                  "a3r_align_create: missing parameter buffers");
    A3R_CHECK_ARG(!s->use_mono || (s->mono && s->shifts), "a3r_align_create: use_mono needs mono and shifts");
    A3R_CHECK_ARG(s->adam_pw_poses && s->adam_depth && s->adam_small, "a3r_align_create: missing Adam state");
    A3R_CHECK_ARG(s->loss_history && s->loss_capacity > 0, "a3r_align_create: missing loss_history");
    size_t off[17];
    const size_t need = ws_layout(s->E, s->N, s->P, off);
    A3R_CHECK_ARG(!s->train_adaptors || s->adam_pw_adaptors, "a3r_align_create: train_adaptors needs adam_pw_adaptors");
    A3R_CHECK_ARG(s->workspace && s->workspace_bytes >= need, "a3r_align_create: workspace too small (%zu < %zu)",
                  s->workspace_bytes, need);
    {
        // the per-chunk partial rows are addressed with 32-bit byte offsets from a buffer resource (store16_wt): both arrays must
        // stay below 2 GiB (config 3: 2 * 4032 * 144 * 64 B = 74 MB)
        const size_t nch = ((size_t)s->P + 1023) / 1024;
        A3R_CHECK_ARG(2 * (size_t)s->E * nch * 64 < (1ull << 31) && (size_t)s->N * nch * 64 < (1ull << 31),
                      "a3r_align_create: E * P / 1024 too large for the 32-bit partial-sum offsets (E=%d N=%d P=%d)", s->E, s->N, s->P);
    }
    // edge indices must be dense 0..N-1 (base_opt.py:164-167)
    std::vector<int> deg(s->N + 1, 0), seen(s->N, 0);
    for (int e = 0; e < s->E; e++) {
        const int i = s->ei_host[e], j = s->ej_host[e];
        A3R_CHECK_ARG(i >= 0 && i < s->N && j >= 0 && j < s->N, "a3r_align_create: bad pair indices (edge %d = %d,%d)", e, i, j);
        deg[i + 1]++; deg[j + 1]++; seen[i] = seen[j] = 1;
    }
    for (int n = 0; n < s->N; n++) A3R_CHECK_ARG(seen[n], "bad pair indices: missing values (image %d has no edge)", n);
    for (int n = 0; n < s->N; n++)
        A3R_CHECK_ARG(deg[n + 1] <= MAX_INC, "a3r_align_create: image %d has %d incident edge sides (limit %d)", n, deg[n + 1], MAX_INC);
    for (int n = 0; n < s->N; n++) {
        A3R_CHECK_ARG(s->imarea_host[n] > 0 && s->imarea_host[n] <= s->P && s->imw_host[n] > 0,
                      "a3r_align_create: bad image shape for image %d", n);
        deg[n + 1] += deg[n];
    }
    std::vector<int> inc(2 * s->E), slot(2 * s->E), fill(s->N, 0);
    for (int e = 0; e < s->E; e++) {
        const int i = s->ei_host[e], j = s->ej_host[e];
        int k = deg[i] + fill[i]++; inc[k] = e * 2 + 0; slot[e * 2 + 0] = k;
        k = deg[j] + fill[j]++;     inc[k] = e * 2 + 1; slot[e * 2 + 1] = k;
    }
    a3r_align_s* a = new (std::nothrow) a3r_align_s();
    A3R_CHECK_ARG(a, "out of host memory");
    char* ws = (char*)s->workspace;
    hipStream_t st = as_stream(stream);
    auto up = [&](size_t o, const void* src, size_t bytes) { return hipMemcpyAsync(ws + o, src, bytes, hipMemcpyHostToDevice, st); };
    hipError_t err = up(off[7], deg.data(), (s->N + 1) * 4);
    if (err == hipSuccess) err = up(off[8], inc.data(), 2 * s->E * 4);
    if (err == hipSuccess) err = up(off[9], slot.data(), 2 * s->E * 4);
    if (err == hipSuccess) err = up(off[10], s->imw_host, s->N * 4);
    if (err == hipSuccess) err = up(off[11], s->imarea_host, s->N * 4);
    // dispatch order of the images: most incident edge sides first (stable: ties keep the image order)
    std::vector<int> order(s->N);
    for (int n = 0; n < s->N; n++) order[n] = n;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deg[x + 1] - deg[x] > deg[y + 1] - deg[y]; });
    std::vector<int> tab((size_t)s->N * 8, 0);
    for (int y = 0; y < s->N; y++) {
        const int n = order[y], kb = deg[n], ke = deg[n + 1];
        tab[y * 8 + 0] = n; tab[y * 8 + 1] = kb; tab[y * 8 + 2] = ke;
        tab[y * 8 + 3] = kb < ke ? inc[kb] : 0;
        tab[y * 8 + 4] = kb + 1 < ke ? inc[kb + 1] : 0;
        tab[y * 8 + 5] = kb + 2 < ke ? inc[kb + 2] : 0;
    }
    if (err == hipSuccess) err = up(off[16], tab.data(), tab.size() * 4);
    if (err == hipSuccess) err = hipMemsetAsync(ws + off[15], 0, (size_t)(s->N + 1) * 4, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);   // host vectors go out of scope
    if (err != hipSuccess) {
        delete a;
        set_error("a3r_align_create: upload failed: %s", hipGetErrorString(err));
        return A3R_EHIP;
    }
    AlignDev& d = a->d;
    d.E = s->E; d.N = s->N; d.P = s->P; d.nchunks = (s->P + CHUNK - 1) / CHUNK;
    d.norm_pw_scale = s->norm_pw_scale; d.train_poses = s->train_poses; d.train_focals = s->train_focals;
    d.train_pp = s->train_pp; d.train_adaptors = s->train_adaptors; d.adam_pw_adaptors = s->adam_pw_adaptors;
    d.base_scale = s->base_scale; d.pw_break = s->pw_break; d.focal_break = s->focal_break;
    d.inv_area_i = (float)(1.0 / s->total_area_i); d.inv_area_j = (float)(1.0 / s->total_area_j);
    d.pred_i = s->pred_i; d.pred_j = s->pred_j; d.w_i = s->w_i; d.w_j = s->w_j;
    d.mono = s->use_mono ? s->mono : nullptr; d.pp0 = s->pp0;
    d.pw_poses = s->pw_poses; d.pw_adaptors = s->pw_adaptors; d.depth = s->depth; d.shifts = s->shifts;
    d.im_poses = s->im_poses; d.im_focals = s->im_focals; d.im_pp = s->im_pp;
    d.adam_pw_poses = s->adam_pw_poses; d.adam_depth = s->adam_depth; d.adam_small = s->adam_small;
    d.edge_xf = (float*)(ws + off[0]); d.img_xf = (float*)(ws + off[1]);
    d.partE = (float*)(ws + off[2]); d.partN = (float*)(ws + off[3]);
    d.gE = (float*)(ws + off[4]); d.gN = (float*)(ws + off[5]); d.lossE = (float*)(ws + off[6]);
    d.gA = (float*)(ws + off[12]);
    d.sumE = (float*)(ws + off[13]); d.sumN = (float*)(ws + off[14]); d.tick = (int*)(ws + off[15]);
    d.order = (const int*)(ws + off[16]);
    {
        // A3R_ALIGN_TAIL=fused: finish the iteration inside the main launch (last-block-done tickets) instead of the two small
        // finalize launches.  Correct and bitwise identical, but measured SLOWER on MI355X (config 2: 147 vs 124 + 18 us per
        // iteration; E = 992: 1005 vs 890 + 46 us): what the tail costs is its serial critical path -- the last image's partial
        // rows, the edge chain rules, the single-block Adam -- not the two launch boundaries, so the
        // separate launches, whose E + N workgroups reduce in parallel, stay the default.
        const char* t = getenv("A3R_ALIGN_TAIL");
        d.fused_tail = (t && !strcmp(t, "fused")) ? 1 : 0;
    }
    d.inc_ptr = (const int*)(ws + off[7]); d.inc = (const int*)(ws + off[8]); d.slot_of = (const int*)(ws + off[9]);
    d.imw = (const int*)(ws + off[10]); d.imarea = (const int*)(ws + off[11]);
    d.loss_history = s->loss_history;
    a->use_mono = s->use_mono != 0; a->dist_l2 = s->dist_l2 != 0; a->steps = 0; a->loss_capacity = s->loss_capacity;
    a->dirty = true;
    a->ei.assign(s->ei_host, s->ei_host + s->E);
    a->ej.assign(s->ej_host, s->ej_host + s->E);
    a->inc = inc;
    *out = a;
    return A3R_OK;
}

extern "C" int a3r_align_destroy(a3r_align_t a) {
    delete a;
    return A3R_OK;
}

template <int MODE>
static void launch_main(a3r_align_s* a, const AdamArgs& ad, float* g_depth, const TailOut& tout, hipStream_t st) {
    dim3 grid(a->d.nchunks, a->d.N), block(TPB);
    // algorithmic bytes of one iteration: 32 B per edge-pixel + 24 B per image-pixel (+4 mono)
    const double bytes = 32.0 * a->d.E * a->d.P + (MODE == 2 ? 24.0 : 4.0) * a->d.N * a->d.P + (a->use_mono ? 4.0 * a->d.N * a->d.P : 0.0);
    ProfScope prof(PK_ALIGN_MAIN, bytes, st);
    const bool vec = a->d.P % 4 == 0;
#define A3R_ALIGN_LAUNCH(MONOV, L2V)                                                                                      \
    do {                                                                                                                 \
        if (vec) hipLaunchKernelGGL((align_main_kernel<MONOV, L2V, MODE, true>), grid, block, 0, st, a->d, ad, g_depth, tout, \
                                    a->d.inc_ptr, a->d.inc, a->d.edge_xf, a->d.img_xf, a->d.imw, a->d.imarea, a->d.order); \
        else hipLaunchKernelGGL((align_main_kernel<MONOV, L2V, MODE, false>), grid, block, 0, st, a->d, ad, g_depth, tout, \
                                a->d.inc_ptr, a->d.inc, a->d.edge_xf, a->d.img_xf, a->d.imw, a->d.imarea, a->d.order);  \
    } while (0)
    if (a->use_mono) {
        if (a->dist_l2) A3R_ALIGN_LAUNCH(true, true); else A3R_ALIGN_LAUNCH(true, false);
    } else {
        if (a->dist_l2) A3R_ALIGN_LAUNCH(false, true); else A3R_ALIGN_LAUNCH(false, false);
    }
#undef A3R_ALIGN_LAUNCH
}

// ---- cloud_opt_flow extras -------------------------------------------------------------------------------------
static size_t flow_ws_layout(int E, int N, int P, size_t* off /*[6]*/) {
    const int nch = (P + CHUNK - 1) / CHUNK;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    off[0] = take((size_t)2 * N * P * 3 * 4);          // gflow
    off[1] = take((size_t)2 * E * nch * NFP * 4);      // partF
    off[2] = take((size_t)2 * E * NFP * 4);            // sumF
    off[3] = take(8 * 4);                              // flow_state
    off[4] = take((size_t)N * 4);                      // lossN
    off[5] = take((size_t)2 * E * 4);                  // other
    return o;
}

extern "C" size_t a3r_align_flow_workspace_bytes(int E, int N, int P) {
    size_t off[6];
    return flow_ws_layout(E, N, P, off);
}

extern "C" int a3r_align_set_flow(a3r_align_t a, const a3r_align_flow_desc* f, void* stream) {
    A3R_CHECK_ARG(a && f, "a3r_align_set_flow: null argument");
    A3R_CHECK_ARG(!a->use_mono, "a3r_align_set_flow: the flow variant has no mono-depth parameterisation (cloud_opt_flow/optimizer.py:52)");
    AlignDev& d = a->d;
    size_t off[6];
    const size_t need = flow_ws_layout(d.E, d.N, d.P, off);
    A3R_CHECK_ARG(f->workspace && f->workspace_bytes >= need, "a3r_align_set_flow: workspace too small (%zu < %zu)", f->workspace_bytes, need);
    A3R_CHECK_ARG(f->temporal_smoothing_weight >= 0.f && f->flow_loss_weight >= 0.f, "a3r_align_set_flow: negative weight");
    if (f->flow_loss_weight > 0.f) {
        A3R_CHECK_ARG(f->flow_ij && f->flow_ji && f->dynamic_mask, "a3r_align_set_flow: flow_loss_weight > 0 needs flow_ij, flow_ji and dynamic_mask");
        A3R_CHECK_ARG(f->H > 0 && f->W > 0 && f->H * f->W == d.P, "a3r_align_set_flow: H*W must equal P (all images of one shape)");
    }
    char* ws = (char*)f->workspace;
    hipStream_t st = as_stream(stream);
    std::vector<int> other(2 * d.E);
    for (int k = 0; k < 2 * d.E; k++) {
        const int e = a->inc[k] >> 1, side = a->inc[k] & 1;
        other[k] = side ? a->ei[e] : a->ej[e];
    }
    A3R_HIP(hipMemcpyAsync(ws + off[5], other.data(), other.size() * 4, hipMemcpyHostToDevice, st));
    A3R_HIP(hipMemsetAsync(ws + off[3], 0, 32, st));
    A3R_HIP(hipMemsetAsync(ws + off[4], 0, (size_t)d.N * 4, st));
    A3R_HIP(hipStreamSynchronize(st));
    d.shared_focal = f->shared_focal; d.tsw = f->temporal_smoothing_weight; d.trans_w = f->translation_weight;
    d.flow_w = f->flow_loss_weight; d.flow_thre = f->flow_loss_thre; d.pxl_thre = f->pxl_thre;
    d.fH = f->H; d.fW = f->W; d.flow_ij = f->flow_ij; d.flow_ji = f->flow_ji; d.dyn = f->dynamic_mask;
    d.gflow = (float*)(ws + off[0]); d.partF = (float*)(ws + off[1]); d.sumF = (float*)(ws + off[2]);
    d.flow_state = (float*)(ws + off[3]); d.lossN = (float*)(ws + off[4]); d.other = (const int*)(ws + off[5]);
    a->flow_start_iter = f->flow_start_iter;
    a->dirty = true;
    return A3R_OK;
}

extern "C" size_t a3r_align_depth_prior_workspace_bytes(int N, int P) {
    return align_up((size_t)N * P * 4, 256) + align_up((size_t)N * 4, 256);
}

extern "C" int a3r_align_set_depth_prior(a3r_align_t a, float weight, const float* init_log_depth, const unsigned char* dynamic_mask,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    A3R_CHECK_ARG(a, "a3r_align_set_depth_prior: null handle");
    A3R_CHECK_ARG(weight >= 0.f, "a3r_align_set_depth_prior: negative weight");
    AlignDev& d = a->d;
    if (weight == 0.f) {
        d.prior_w = 0.f; d.prior_init = nullptr; d.prior_dyn = nullptr; d.gprior = nullptr; d.lossP = nullptr;
        return A3R_OK;
    }
    A3R_CHECK_ARG(!a->use_mono, "a3r_align_set_depth_prior: the flow variant has no mono-depth parameterisation");
    A3R_CHECK_ARG(init_log_depth, "a3r_align_set_depth_prior: the initial depth maps are missing (_set_init_depthmap has not run)");
    const size_t need = a3r_align_depth_prior_workspace_bytes(d.N, d.P);
    A3R_CHECK_ARG(workspace && workspace_bytes >= need, "a3r_align_set_depth_prior: workspace too small (%zu < %zu)", workspace_bytes, need);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "a3r_align_set_depth_prior: workspace must be 16-byte aligned");
    d.prior_w = weight; d.prior_init = init_log_depth; d.prior_dyn = dynamic_mask;
    d.gprior = static_cast<float*>(workspace);
    d.lossP = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)d.N * d.P * 4, 256));
    (void)stream;
    return A3R_OK;
}

// the ego-flow pass of one iteration (before the main kernel): unscaled sums, then the normalisers / drop decision
static void launch_flow(a3r_align_s* a, int epoch, hipStream_t st) {
    AlignDev& d = a->d;
    if (d.prior_w > 0.f) hipLaunchKernelGGL(align_depth_prior_kernel, dim3(d.N), dim3(PRIOR_TPB), 0, st, d);
    d.flow_on = (d.flow_w > 0.f && epoch >= a->flow_start_iter) ? 1 : 0;
    if (!d.flow_on) return;
    ProfScope prof(PK_ALIGN_SMALL, 0.0, st);
    static const bool flow_v1 = getenv("A3R_ALIGN_FLOW") && std::string(getenv("A3R_ALIGN_FLOW")) == "v1";    // A/B switch: the first form
    const bool vec = d.P % 4 == 0 && !flow_v1 && ((reinterpret_cast<uintptr_t>(d.flow_ij) | reinterpret_cast<uintptr_t>(d.flow_ji) |
                                                   reinterpret_cast<uintptr_t>(d.gflow) | reinterpret_cast<uintptr_t>(d.dyn)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(align_flow_vec_kernel, dim3(d.nchunks, d.N), dim3(TPB), 0, st, d, d.inc_ptr, d.inc, d.other, d.img_xf);
    else hipLaunchKernelGGL(align_flow_kernel, dim3(d.nchunks, d.N), dim3(TPB), 0, st, d, d.inc_ptr, d.inc, d.other, d.img_xf);
    hipLaunchKernelGGL(align_flow_reduce_kernel, dim3(2 * d.E), dim3(64), 0, st, d);
    hipLaunchKernelGGL(align_flow_decide_kernel, dim3(1), dim3(TPB), 0, st, d, d.inc);
}

extern "C" int a3r_align_step_epoch(a3r_align_t a, float lr, int epoch, void* stream) {
    A3R_CHECK_ARG(a, "a3r_align_step: null handle");
    A3R_CHECK_ARG(a->steps < a->loss_capacity, "a3r_align_step: loss_history full (%d)", a->loss_capacity);
    hipStream_t st = as_stream(stream);
    const int t = a->steps + 1;
    AdamArgs ad;
    ad.lr = lr;
    ad.step_size = (float)((double)lr / (1.0 - pow((double)ADAM_B1, t)));
    ad.bc2_sqrt = (float)sqrt(1.0 - pow((double)ADAM_B2, t));
    ad.step = a->steps;
    refresh_if_dirty(a, st);
    launch_flow(a, epoch, st);
    const TailOut tout = {nullptr, nullptr, nullptr, nullptr};
    launch_main<2>(a, ad, nullptr, tout, st);
    if (!a->d.fused_tail) {
        ProfScope prof(PK_ALIGN_SMALL, 0.0, st);
        hipLaunchKernelGGL(align_finalize_a_kernel, dim3(a->d.E + a->d.N), dim3(64), 0, st, a->d, 0);
        hipLaunchKernelGGL((align_finalize_b_kernel<2>), dim3(1), dim3(TPB), 0, st, a->d, ad, tout);
    }
    A3R_LAUNCH_CHECK();
    a->steps++;
    return A3R_OK;
}

extern "C" int a3r_align_run(a3r_align_t a, const float* lrs_host, int n, int first_epoch, void* stream) {
    // n iterations with the learning rates lrs_host[0..n) (the caller evaluates its schedule, base_opt.py:451-457) enqueued from one
    // native loop: three launches per iteration and nothing else on the host -- a Python loop around a3r_align_step is launch-bound
    // on a busy host (2.8 k instead of 7 k iterations/s were measured on one box)
    A3R_CHECK_ARG(a && lrs_host && n >= 0, "a3r_align_run: bad argument");
    A3R_CHECK_ARG(a->steps + n <= a->loss_capacity, "a3r_align_run: loss_history too small (%d + %d > %d)", a->steps, n, a->loss_capacity);
    for (int k = 0; k < n; k++)
        if (int rc = a3r_align_step_epoch(a, lrs_host[k], first_epoch + k, stream)) return rc;
    return A3R_OK;
}

extern "C" int a3r_align_step(a3r_align_t a, float lr, void* stream) {
    return a3r_align_step_epoch(a, lr, a ? a->steps : 0, stream);     // epoch = iteration index of this handle
}

extern "C" int a3r_align_loss(a3r_align_t a, float* loss_dev, void* stream) {
    A3R_CHECK_ARG(a && loss_dev, "a3r_align_loss: null argument");
    hipStream_t st = as_stream(stream);
    AdamArgs ad = {};
    refresh_if_dirty(a, st);
    launch_flow(a, 1 << 30, st);                                       // net() defaults to epoch=9999: flow term active
    const TailOut tout = {nullptr, nullptr, loss_dev, nullptr};
    launch_main<0>(a, ad, nullptr, tout, st);
    if (!a->d.fused_tail) {
        hipLaunchKernelGGL(align_finalize_a_kernel, dim3(a->d.E + a->d.N), dim3(64), 0, st, a->d, 1);
        hipLaunchKernelGGL((align_finalize_b_kernel<0>), dim3(1), dim3(TPB), 0, st, a->d, ad, tout);
    }
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_align_grad_full(a3r_align_t a, int epoch, float* g_pw_poses, float* g_pw_adaptors, float* g_depth, float* g_small,
                                   float* loss_dev, void* stream) {
    A3R_CHECK_ARG(a && g_pw_poses && g_depth && g_small && loss_dev, "a3r_align_grad: null argument");
    hipStream_t st = as_stream(stream);
    AdamArgs ad = {};
    refresh_if_dirty(a, st);
    launch_flow(a, epoch, st);
    const TailOut tout = {g_pw_poses, g_small, loss_dev, g_pw_adaptors};
    launch_main<1>(a, ad, g_depth, tout, st);
    if (!a->d.fused_tail) {
        hipLaunchKernelGGL(align_finalize_a_kernel, dim3(a->d.E + a->d.N), dim3(64), 0, st, a->d, 0);
        hipLaunchKernelGGL((align_finalize_b_kernel<1>), dim3(1), dim3(TPB), 0, st, a->d, ad, tout);
    }
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_align_grad_epoch(a3r_align_t a, int epoch, float* g_pw_poses, float* g_depth, float* g_small, float* loss_dev,
                                    void* stream) {
    return a3r_align_grad_full(a, epoch, g_pw_poses, nullptr, g_depth, g_small, loss_dev, stream);
}

extern "C" int a3r_align_grad(a3r_align_t a, float* g_pw_poses, float* g_depth, float* g_small, float* loss_dev,
                              void* stream) {
    return a3r_align_grad_epoch(a, 1 << 30, g_pw_poses, g_depth, g_small, loss_dev, stream);
}

extern "C" int a3r_align_flow_state(a3r_align_t a, float* state_host5) {
    A3R_CHECK_ARG(a && state_host5, "a3r_align_flow_state: null argument");
    A3R_CHECK_ARG(a->d.flow_state, "a3r_align_flow_state: a3r_align_set_flow has not been called");
    A3R_HIP(hipMemcpy(state_host5, a->d.flow_state, 5 * sizeof(float), hipMemcpyDeviceToHost));
    return A3R_OK;
}

extern "C" int a3r_align_invalidate(a3r_align_t a) {
    A3R_CHECK_ARG(a, "a3r_align_invalidate: null handle");
    a->dirty = true;
    return A3R_OK;
}

#ifdef A3R_ALIGN_STAMPS
extern "C" int a3r_debug_align_stamps(unsigned long long* host, int n_words) {
    A3R_HIP(hipDeviceSynchronize());
    A3R_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_align_stamps), (size_t)n_words * 8));
    return A3R_OK;
}
#endif

extern "C" int a3r_align_steps_done(a3r_align_t a) { return a ? a->steps : -1; }

extern "C" int a3r_align_pose_matrices(a3r_align_t a, float* edge_M, float* img_R, void* stream) {
    A3R_CHECK_ARG(a && edge_M && img_R, "a3r_align_pose_matrices: null argument");
    hipStream_t st = as_stream(stream);
    a->dirty = true;
    refresh_if_dirty(a, st);
    const int n = (a->d.E > a->d.N ? a->d.E : a->d.N) * 12;
    hipLaunchKernelGGL(align_export_xf_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a->d, edge_M, img_R);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
