// HBM-bound kernels of the pair forward: LayerNorm, standalone RoPE-2D, patchify, bilinear x2,
// final 1x1 conv + postprocess, weight repacking.  All fp32, 16-byte vector accesses where the
// layout allows, one wave per row for the row reductions (64-wide DPP/shuffle sums).
#include "common.h"
#include "bf3.h"
#include "fh2.h"
#include <cmath>
#include <cstdlib>

namespace a3r {

// ------------------------------------------------------------------------------------------- LayerNorm
// nn.LayerNorm(D, eps) (croco.py:34; blocks.py:118-123,180-185): one wave per row, row kept in registers.
// BF3: write the normalised row in bf3 form (bf3.h) instead of fp32 -- the input format of gemm_bf3.hip.
template <int VPL, bool BF3>   // float4 per lane: D = 256 * VPL
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int M,
                                                         int D, float eps, int pair) {
#pragma clang fp contract(off)      // the fp32 and the bf3 instantiation must round identically (bf3 output == split of the fp32 one)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        v[i] = xr[lane + 64 * i];
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        v[i] = v[i] - mean;
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)D + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y + (size_t)row * D);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w);
    const f32x4* br = reinterpret_cast<const f32x4*>(b);
    char* y3 = reinterpret_cast<char*>(y) + bf3_row_offset(row, D, pair);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        // explicit fma: the fp32 and the bf3 instantiation must round identically (bf3 output == split of the fp32 output, tested)
        const f32x4 t = v[i] * rstd, wv = wr[lane + 64 * i], bv = br[lane + 64 * i];
        const f32x4 o = {__builtin_fmaf(t.x, wv.x, bv.x), __builtin_fmaf(t.y, wv.y, bv.y), __builtin_fmaf(t.z, wv.z, bv.z),
                         __builtin_fmaf(t.w, wv.w, bv.w)};
        if (BF3) bf3_store4(y3, (lane + 64 * i) * 4, o, pair);
        else yr[lane + 64 * i] = o;
    }
}

// Row-PAIR bf3 output (the layout every transformer GEMM input uses, bf3.h): ONE WAVE NORMALISES TWO ROWS (2j, 2j + 1) and
// stores their interleaved image -- 12 D contiguous bytes -- through LDS as whole 1 KB wave stores.  The one-row form above writes
// a row's planes as 8-byte pieces, 16 of every 48 bytes per instruction (a lane owns 4 consecutive k = half of each plane's
// 16-byte unit), and in the pair layout a single row is 192-byte runs at a 384-byte stride; here every store instruction covers
// eight whole cache lines, and the two rows' loads and reductions are in flight together.  Same arithmetic, same rounding.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_pair_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, char* __restrict__ y3, int M, int D, float eps) {
#pragma clang fp contract(off)
    // per wave two 3 KB images: the pair's 256 k of one step i (8 k-blocks x 384 B, contiguous in the pair layout), double-buffered
    __shared__ __attribute__((aligned(16))) char img_all[4][2][3072];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + wave) * 2;
    if (r0 >= M) return;                                                        // wave-uniform; M is even in this layout's callers
    const bool two = r0 + 1 < M;
    const f32x4* x0 = reinterpret_cast<const f32x4*>(x + (size_t)r0 * D);
    const f32x4* x1 = reinterpret_cast<const f32x4*>(x + (size_t)(two ? r0 + 1 : r0) * D);
    f32x4 v0[VPL], v1[VPL];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        v0[i] = x0[lane + 64 * i];
        v1[i] = x1[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        s0 += v0[i].x + v0[i].y + v0[i].z + v0[i].w;
        s1 += v1[i].x + v1[i].y + v1[i].z + v1[i].w;
    }
    const float mean0 = wave_sum(s0) / (float)D, mean1 = wave_sum(s1) / (float)D;
    float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        v0[i] = v0[i] - mean0;
        v1[i] = v1[i] - mean1;
        ss0 += v0[i].x * v0[i].x + v0[i].y * v0[i].y + v0[i].z * v0[i].z + v0[i].w * v0[i].w;
        ss1 += v1[i].x * v1[i].x + v1[i].y * v1[i].y + v1[i].z * v1[i].z + v1[i].w * v1[i].w;
    }
    const float rstd0 = 1.f / sqrtf(wave_sum(ss0) / (float)D + eps), rstd1 = 1.f / sqrtf(wave_sum(ss1) / (float)D + eps);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w);
    const f32x4* br = reinterpret_cast<const f32x4*>(b);
    char* dst = y3 + (size_t)(r0 >> 1) * ((size_t)12 * D);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        char* img = img_all[wave][i & 1];
        const f32x4 wv = wr[lane + 64 * i], bv = br[lane + 64 * i];
        const f32x4 t0 = v0[i] * rstd0, t1 = v1[i] * rstd1;
        const f32x4 o0 = {__builtin_fmaf(t0.x, wv.x, bv.x), __builtin_fmaf(t0.y, wv.y, bv.y), __builtin_fmaf(t0.z, wv.z, bv.z),
                          __builtin_fmaf(t0.w, wv.w, bv.w)};
        const f32x4 o1 = {__builtin_fmaf(t1.x, wv.x, bv.x), __builtin_fmaf(t1.y, wv.y, bv.y), __builtin_fmaf(t1.z, wv.z, bv.z),
                          __builtin_fmaf(t1.w, wv.w, bv.w)};
        bf3_store4(img, lane * 4, o0, 1);                                      // row parity 0: offset 0 of every 384-byte k block
        bf3_store4(img + 192, lane * 4, o1, 1);                                // row parity 1: + 192
        __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): the image is wave-private
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int un = u * 64 + lane;
            const u32x4 dv = *reinterpret_cast<const u32x4*>(img + un * 16);
            if (two || ((un * 16) % 384) < 192) *reinterpret_cast<u32x4*>(dst + (size_t)i * 3072 + (size_t)un * 16) = dv;
        }
    }
}

// fh2 output (fh2.h; scale 1): one wave per row, a lane owns 8 CONSECUTIVE k per step (two adjacent float4 loads), so it writes the
// two planes' 16-byte units of its group -- 32 contiguous bytes per lane, 2 KB contiguous per wave and step.
template <int VPL, bool STATS>   // 8-k groups per lane: D <= 512 VPL (D % 8 == 0); a lane whose group lies past the row contributes zeros and stores nothing
__global__ __launch_bounds__(256) void layernorm_fh2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, char* __restrict__ y2, int M, int D, float eps,
                                                             float scale, unsigned* __restrict__ absmax) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int G = D >> 3;                                      // groups of 8 consecutive k in a row
    float amax = 0.f;
    // persistent rows loop (the grid is capped): the range statistics cost one atomic per workgroup, not one per row
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    f32x4 v[VPL][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const bool in = lane + 64 * i < G;
        v[i][0] = in ? xr[2 * (lane + 64 * i)] : f32x4{0.f, 0.f, 0.f, 0.f};
        v[i][1] = in ? xr[2 * (lane + 64 * i) + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0].x + v[i][0].y + v[i][0].z + v[i][0].w) + (v[i][1].x + v[i][1].y + v[i][1].z + v[i][1].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const bool in = lane + 64 * i < G;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            v[i][h] = v[i][h] - mean;
            if (in) ss += v[i][h].x * v[i][h].x + v[i][h].y * v[i][h].y + v[i][h].z * v[i][h].z + v[i][h].w * v[i][h].w;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)D + eps);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w);
    const f32x4* br = reinterpret_cast<const f32x4*>(b);
    char* yr = y2 + (size_t)row * fh2_row_bytes(D);
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        if (lane + 64 * i >= G) continue;
        f32x4 o[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = 2 * (lane + 64 * i) + h;
            const f32x4 t = v[i][h] * rstd, wv = wr[c], bv = br[c];
            o[h] = f32x4{__builtin_fmaf(t.x, wv.x, bv.x), __builtin_fmaf(t.y, wv.y, bv.y), __builtin_fmaf(t.z, wv.z, bv.z),
                         __builtin_fmaf(t.w, wv.w, bv.w)} * scale;
            if (STATS) amax = fh2_amax4(amax, o[h]);
        }
        fh2_store8(yr, (lane + 64 * i) * 8, o[0], o[1]);
    }
    }
    if (STATS) {
        __shared__ unsigned s_red[4];
        fh2_publish_block(absmax, amax, s_red);
    }
}
// any D % 32 == 0: lanes stride over the row's 8-k groups (three passes over an L1/L2-resident row)
__global__ __launch_bounds__(256) void layernorm_fh2_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ b, char* __restrict__ y2, int M, int D, float eps,
                                                                     float scale, unsigned* __restrict__ absmax) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    float amax = 0.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float c = xr[i] - mean; ss += c * c; }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)D + eps);
    char* yr = y2 + (size_t)row * fh2_row_bytes(D);
    for (int k0 = lane * 8; k0 < D; k0 += 512) {
        f32x4 o[2];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j >> 2][j & 3] = __builtin_fmaf((xr[k0 + j] - mean) * rstd, w[k0 + j], b[k0 + j]) * scale;
        amax = fh2_amax4(fh2_amax4(amax, o[0]), o[1]);
        fh2_store8(yr, k0, o[0], o[1]);
    }
    }
    __shared__ unsigned s_red[4];
    fh2_publish_block(absmax, amax, s_red);
}

// generic fallback (any D % 4 == 0; D % 8 == 0 for BF3): one wave per row, three passes over an L1/L2-resident row
template <bool BF3>
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ y, int M,
                                                                 int D, float eps, int pair) {
#pragma clang fp contract(off)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float c = xr[i] - mean; ss += c * c; }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)D + eps);
    if (BF3) {
        char* y3 = reinterpret_cast<char*>(y) + bf3_row_offset(row, D, pair);
        for (int i = lane * 4; i < D; i += 256) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = __builtin_fmaf((xr[i + j] - mean) * rstd, w[i + j], b[i + j]);
            bf3_store4(y3, i, o, pair);
        }
    } else {
        float* yr = y + (size_t)row * D;
        for (int i = lane; i < D; i += 64) yr[i] = __builtin_fmaf((xr[i] - mean) * rstd, w[i], b[i]);
    }
}

// ------------------------------------------------------------------------------------------- RoPE-2D (standalone)
// curope.rope_2d semantics (curope.cpp:11-47, kernels.cu:17-82): tokens [B,N,H,D] in place.
__global__ void rope2d_kernel(float* tok, const int64_t* pos, int B, int N, int H, int D, float base, float fwd) {
    const int Q = D / 4;
    const long total = (long)B * N * H * 2 * Q;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        long r = i / Q;
        const int xy = (int)(r % 2); r /= 2;
        const int hh = (int)(r % H); r /= H;      // r = b*N + n
        const float p = (float)pos[r * 2 + xy];
        const float ang = fwd * p / powf(base, (float)q / (float)Q);
        const float c = cosf(ang), s = sinf(ang);
        float* t = tok + (r * H + hh) * (long)D + xy * 2 * Q + q;
        const float u = t[0], v = t[Q];
        t[0] = u * c - v * s;
        t[Q] = v * c + u * s;
    }
}

// ------------------------------------------------------------------------------------------- patchify
// cols[(b*nh + ty)*nw + tx][c*256 + py*16 + px] = img(b, c, ty*16+py, tx*16+px)
__global__ void patchify_kernel(const float* __restrict__ img, float* __restrict__ cols, int B, int C, int H, int W,
                                long sb, long sc, long sy, long sx) {
    const int nh = H / 16, nw = W / 16, K = C * 256;
    const long total = (long)B * nh * nw * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long t = i / K;
        const int tx = (int)(t % nw), ty = (int)((t / nw) % nh), b = (int)(t / ((long)nw * nh));
        const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
        cols[i] = img[b * sb + c * sc + (long)(ty * 16 + py) * sy + (long)(tx * 16 + px) * sx];
    }
}

// ------------------------------------------------------------------------------------------- bilinear x2, align_corners=True
// Index arithmetic in fp32 exactly as ATen's upsample_bilinear2d (scale = (in-1)/(out-1); src = scale*dst).
// BF3: write the result in bf3 form (rows = output pixels, K = C) -- the input of the following 1x1 / 3x3 conv on the bf3 kernel.
// one fixed rounding order for the interpolation in every output format (products and fused multiply-adds spelled out: left to
// contraction the fp32 and the fh2 kernels would round differently and "fh2 output == split of the fp32 output" would not hold)
__device__ __forceinline__ f32x4 bilerp4(f32x4 v00, f32x4 v01, f32x4 v10, f32x4 v11, float lx0, float lx1, float ly0, float ly1) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float t0 = __fmaf_rn(v01[e], lx1, __fmul_rn(v00[e], lx0));
        const float t1 = __fmaf_rn(v11[e], lx1, __fmul_rn(v10[e], lx0));
        o[e] = __fmaf_rn(t1, ly1, __fmul_rn(t0, ly0));
    }
    return o;
}

// FMT 0: fp32, 1: bf3, 2: fh2 output.  Grid: x over (output column, channel group) of one output row, y over (batch, output row):
// the row / batch decomposition is scalar work once per workgroup and a thread does ONE 32-bit division (the first version took
// three 64-bit divisions by runtime values per thread).  A thread produces 4 channels (fp32 / bf3) or 8 channels (fh2: the two
// planes' 16-byte units of one 8-k group = 32 contiguous bytes, and half the index arithmetic per element -- PMC showed the
// 4-channel fh2 form VALU-bound at 150 instructions per thread).
template <int FMT>
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4,
                                                         int Hc, int Wc, float scale, unsigned* __restrict__ absmax) {
    constexpr int Q = FMT == 2 ? 2 : 1;                        // float4 per thread
    const float sh = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float sw = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x);
    f32x4* yv = reinterpret_cast<f32x4*>(y);
    const unsigned CG = (unsigned)C4 / Q;                      // channel groups per pixel
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned ox = idx / CG, c = (idx - ox * CG) * Q;     // c: first float4 of this thread
    const bool live = ox < (unsigned)Wc;                     // (no early return in the fh2 form: the range statistics need every lane at the end)
    if (FMT != 2 && !live) return;
    const float fx = sw * (float)ox;
    int x0 = (int)fx;
    x0 = x0 < W - 1 ? x0 : W - 1;
    const int x1 = x0 < W - 1 ? x0 + 1 : x0;
    const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    float amax = 0.f;
    for (int row = blockIdx.y; live && row < B * Hc; row += gridDim.y) {   // (batch, output row): uniform per workgroup
        const int b = row / Hc, oy = row - b * Hc;
        const float fy = sh * (float)oy;
        int y0 = (int)fy;
        y0 = y0 < H - 1 ? y0 : H - 1;
        const int y1 = y0 < H - 1 ? y0 + 1 : y0;
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
        const long rb = (long)b * H;
        const f32x4* r0 = xv + (rb + y0) * W * C4 + c;
        const f32x4* r1 = xv + (rb + y1) * W * C4 + c;
        f32x4 o[Q];
#pragma unroll
        for (int q = 0; q < Q; q++)
            o[q] = bilerp4(r0[(long)x0 * C4 + q], r0[(long)x1 * C4 + q], r1[(long)x0 * C4 + q], r1[(long)x1 * C4 + q], lx0, lx1, ly0, ly1);
        const long pix = (long)row * Wc + ox;
        if (FMT == 1) bf3_store4(reinterpret_cast<char*>(y) + pix * ((size_t)C4 * 24), c * 4, o[0]);
        else if (FMT == 2) {
            const f32x4 lo = o[0] * scale, hi = o[Q - 1] * scale;
            amax = fh2_amax4(fh2_amax4(amax, lo), hi);
            fh2_store8(reinterpret_cast<char*>(y) + pix * ((size_t)C4 * 16), c * 4, lo, hi);
        } else yv[pix * C4 + c] = o[0];
    }
    if (FMT == 2) {
        __shared__ unsigned s_red[4];
        fh2_publish_block(absmax, amax, s_red);
    }
}

// ------------------------------------------------------------------------------------------- head final
// conv1x1 C -> 4 (+bias) then postprocess (postprocess.py:10-58). 32 lanes per pixel (C == 128: one float4 each).
__global__ __launch_bounds__(256) void head_final_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ pts,
                                                          float* __restrict__ conf, long P, int C) {
    const int lane = threadIdx.x & 63, sub = lane & 31;
    const long pix0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const long stride = (long)gridDim.x * 8;
    const long iters = (P + stride - 1) / stride;      // uniform trip count: the shuffles need every lane
    const int nvec = C / 4;
    for (long it = 0; it < iters; it++) {
        const long pix = pix0 + it * stride;
        const bool ok = pix < P;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const f32x4* xr = reinterpret_cast<const f32x4*>(x + pix * C);
            for (int j = sub; j < nvec; j += 32) {
                const f32x4 v = xr[j];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const f32x4 ww = reinterpret_cast<const f32x4*>(w + o * C)[j];
                    acc[o] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) acc[o] += __shfl_xor(acc[o], m);
            acc[o] += bias[o];
        }
        if (ok && sub == 0) {
            // reference order: xyz / d.clip(1e-8) * expm1(d)   (postprocess.py:37-46)
            const float d = sqrtf(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
            const float dd = fmaxf(d, 1e-8f), em = expm1f(d);
            pts[pix * 3 + 0] = acc[0] / dd * em;
            pts[pix * 3 + 1] = acc[1] / dd * em;
            pts[pix * 3 + 2] = acc[2] / dd * em;
            conf[pix] = 1.f + expf(acc[3]);
        }
    }
}

// C == 128 (the DPT head's last_dim): 8 lanes per pixel, a lane owns 16 consecutive channels = 64 contiguous bytes, so a wave reads
// 8 pixels = 4 KB contiguous per step (the 32-lanes-per-pixel form above has 1 KB per step in flight and re-loads its four weight
// vectors every pixel); the 4 x 16 weights of a lane stay in registers; three exchange steps finish the four dot products.
__global__ __launch_bounds__(256) void head_final128_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ pts,
                                                             float* __restrict__ conf, long P) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    f32x4 wr[4][4];
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int q = 0; q < 4; q++) wr[o][q] = reinterpret_cast<const f32x4*>(w + o * 128)[sub * 4 + q];
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const long iters = (P + nwaves * 8 - 1) / (nwaves * 8);       // uniform trip count: the exchanges need every lane
    for (long it = 0; it < iters; it++) {
        const long pix = (it * nwaves + wave) * 8 + (lane >> 3);
        const bool ok = pix < P;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const f32x4* xr = reinterpret_cast<const f32x4*>(x + pix * 128) + sub * 4;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = xr[q];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[o] += v[q].x * wr[o][q].x + v[q].y * wr[o][q].y + v[q].z * wr[o][q].z + v[q].w * wr[o][q].w;
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            acc[o] += __shfl_xor(acc[o], 1);
            acc[o] += __shfl_xor(acc[o], 2);
            acc[o] += __shfl_xor(acc[o], 4);
        }
        if (ok && sub == 0) {
            const float a0 = acc[0] + b0, a1 = acc[1] + b1, a2 = acc[2] + b2, a3 = acc[3] + b3;
            // reference order: xyz / d.clip(1e-8) * expm1(d)   (postprocess.py:37-46)
            const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
            const float dd = fmaxf(d, 1e-8f), em = expm1f(d);
            pts[pix * 3 + 0] = a0 / dd * em;
            pts[pix * 3 + 1] = a1 / dd * em;
            pts[pix * 3 + 2] = a2 / dd * em;
            conf[pix] = 1.f + expf(a3);
        }
    }
}

// ------------------------------------------------------------------------------------------- weight repacking
// [Cout, Cin, 3, 3] -> [Cout, 3, 3, Cin]
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
    const long total = (long)Cout * Cin * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int tap = (int)((i / Cin) % 9);
        const int co = (int)(i / ((long)Cin * 9));
        wp[i] = w[((long)co * Cin + ci) * 9 + tap];
    }
}
// [Cin, Cout, s, s] -> [(dy*s+dx)*Cout + co][Cin]
__global__ void pack_convT_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int s) {
    const long total = (long)Cin * Cout * s * s;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const long n = i / Cin;
        const int co = (int)(n % Cout), tap = (int)(n / Cout);
        wp[i] = w[((long)ci * Cout + co) * (s * s) + tap];
    }
}

static inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// ------------------------------------------------------------------------------------------- weighted Umeyama moments
// The raw second moments of B weighted point-set pairs (x_b, y_b, w_b), P points each, for the similarity registrations of the
// aligner's initialisation (cloud_opt/init_im_poses.py:415-418: s R x + T ~ y): per problem and 1024-point chunk 17 doubles
//   [0] sum w   [1..3] sum w x   [4..6] sum w y   [7] sum w |x|^2   [8..16] sum w y_r x_c (r major)
// One launch for all problems (e.g. every edge of the pair graph); fixed summation order, no atomics.
__global__ __launch_bounds__(256) void umeyama_moments_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ w, const long* __restrict__ xo,
                                                               const long* __restrict__ yo, const long* __restrict__ wo, int P,
                                                               double* __restrict__ partial) {
    __shared__ double red[4][17];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + xo[b];
    const float* yb = y + yo[b];
    const float* wb = w + wo[b];
    double a[17];
#pragma unroll
    for (int j = 0; j < 17; j++) a[j] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = chunk * 1024 + i * 256 + tid;
        if (p < P) {
            const double ww = wb[p];
            const double x0 = xb[3 * p], x1 = xb[3 * p + 1], x2 = xb[3 * p + 2];
            const double y0 = yb[3 * p], y1 = yb[3 * p + 1], y2 = yb[3 * p + 2];
            a[0] += ww;
            a[1] += ww * x0; a[2] += ww * x1; a[3] += ww * x2;
            a[4] += ww * y0; a[5] += ww * y1; a[6] += ww * y2;
            a[7] += ww * (x0 * x0 + x1 * x1 + x2 * x2);
            a[8] += ww * y0 * x0; a[9] += ww * y0 * x1; a[10] += ww * y0 * x2;
            a[11] += ww * y1 * x0; a[12] += ww * y1 * x1; a[13] += ww * y1 * x2;
            a[14] += ww * y2 * x0; a[15] += ww * y2 * x1; a[16] += ww * y2 * x2;
        }
    }
#pragma unroll
    for (int j = 0; j < 17; j++) {
        double v = a[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (tid < 17) partial[((size_t)b * gridDim.x + chunk) * 17 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

}  // namespace a3r
using namespace a3r;

extern "C" int a3r_umeyama_chunks(int P) { return P > 0 ? (P + 1023) / 1024 : 0; }

extern "C" int a3r_umeyama_moments(const float* x, const float* y, const float* w, const long* x_off, const long* y_off,
                                   const long* w_off, int B, int P, double* partial, void* stream) {
    A3R_CHECK_ARG(x && y && w && x_off && y_off && w_off && partial, "a3r_umeyama_moments: null pointer");
    A3R_CHECK_ARG(B > 0 && B <= 65535 && P > 0, "a3r_umeyama_moments: bad shape B=%d P=%d", B, P);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(umeyama_moments_kernel, dim3(a3r_umeyama_chunks(P), B), dim3(256), 0, st, x, y, w, x_off, y_off, w_off, P, partial);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

template <bool BF3>
static int launch_layernorm(const float* x, const float* w, const float* b, float* y, int M, int D, float eps, void* stream, int pair = 0) {
    hipStream_t st = as_stream(stream);
    ProfScope prof(PK_LAYERNORM, (BF3 ? 10.0 : 8.0) * M * D, st);
    dim3 grid((M + 3) / 4), block(256);
    static const bool one_row = getenv("A3R_LN_ONE_ROW") != nullptr;           // developer A/B switch
    if (BF3 && pair && !one_row && (D == 1024 || D == 768 || D == 256)) {
        const dim3 g2((M + 7) / 8);
        char* y3 = reinterpret_cast<char*>(y);
        if (D == 1024) hipLaunchKernelGGL(layernorm_pair_kernel<4>, g2, block, 0, st, x, w, b, y3, M, D, eps);
        else if (D == 768) hipLaunchKernelGGL(layernorm_pair_kernel<3>, g2, block, 0, st, x, w, b, y3, M, D, eps);
        else hipLaunchKernelGGL(layernorm_pair_kernel<1>, g2, block, 0, st, x, w, b, y3, M, D, eps);
        A3R_LAUNCH_CHECK();
        return A3R_OK;
    }
    if (D == 1024) hipLaunchKernelGGL((layernorm_kernel<4, BF3>), grid, block, 0, st, x, w, b, y, M, D, eps, pair);
    else if (D == 768) hipLaunchKernelGGL((layernorm_kernel<3, BF3>), grid, block, 0, st, x, w, b, y, M, D, eps, pair);
    else if (D == 256) hipLaunchKernelGGL((layernorm_kernel<1, BF3>), grid, block, 0, st, x, w, b, y, M, D, eps, pair);
    else hipLaunchKernelGGL(layernorm_generic_kernel<BF3>, grid, block, 0, st, x, w, b, y, M, D, eps, pair);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_layernorm(const float* x, const float* w, const float* b, float* y, int M, int D, float eps,
                             void* stream) {
    A3R_CHECK_ARG(x && w && b && y, "a3r_layernorm: null pointer");
    A3R_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0, "a3r_layernorm: bad shape M=%d D=%d", M, D);
    return launch_layernorm<false>(x, w, b, y, M, D, eps, stream);
}

extern "C" int a3r_layernorm_fh2(const float* x, const float* w, const float* b, void* y2, int M, int D, float eps, float scale,
                                 unsigned* absmax, void* stream) {
    A3R_CHECK_ARG(x && w && b && y2, "a3r_layernorm_fh2: null pointer");
    A3R_CHECK_ARG(scale > 0.f && std::isfinite(scale), "a3r_layernorm_fh2: scale must be positive and finite");
    A3R_CHECK_ARG(M > 0 && D > 0 && D % 32 == 0, "a3r_layernorm_fh2: bad shape M=%d D=%d (D must be a multiple of 32)", M, D);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(y2) & 15) == 0, "a3r_layernorm_fh2: y2 must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PK_LAYERNORM, 8.0 * M * D, st);
    const int nblk = (M + 3) / 4;
    // with statistics: a persistent rows loop (16 workgroups per CU, one atomic each); without: one row per wave
    dim3 grid(absmax && nblk > 4096 ? 4096 : nblk), block(256);
    char* y = static_cast<char*>(y2);
    // rows up to 1536 wide stay in registers (one, two or three 8-k groups per lane; D = 768 uses two with the upper lanes idle in
    // the second: 2.7 -> 4.5+ TB/s against the three-pass generic kernel it used before)
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    if (aligned && D <= 512) { if (absmax) hipLaunchKernelGGL((layernorm_fh2_kernel<1, true>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); else hipLaunchKernelGGL((layernorm_fh2_kernel<1, false>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); }
    else if (aligned && D <= 1024) { if (absmax) hipLaunchKernelGGL((layernorm_fh2_kernel<2, true>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); else hipLaunchKernelGGL((layernorm_fh2_kernel<2, false>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); }
    else if (aligned && D <= 1536) { if (absmax) hipLaunchKernelGGL((layernorm_fh2_kernel<3, true>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); else hipLaunchKernelGGL((layernorm_fh2_kernel<3, false>), grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax); }
    else hipLaunchKernelGGL(layernorm_fh2_generic_kernel, grid, block, 0, st, x, w, b, y, M, D, eps, scale, absmax);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_layernorm_bf3(const float* x, const float* w, const float* b, void* y3, int M, int D, float eps, int pair,
                                 void* stream) {
    A3R_CHECK_ARG(x && w && b && y3, "a3r_layernorm_bf3: null pointer");
    A3R_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "a3r_layernorm_bf3: bad shape M=%d D=%d (D must be a multiple of 8)", M, D);
    A3R_CHECK_ARG(!pair || D % 32 == 0, "a3r_layernorm_bf3: the row-pair layout needs D (%d) to be a multiple of 32", D);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(y3) & 15) == 0, "a3r_layernorm_bf3: y3 must be 16-byte aligned");
    return launch_layernorm<true>(x, w, b, static_cast<float*>(y3), M, D, eps, stream, pair ? 1 : 0);
}

extern "C" int a3r_rope2d(float* tokens, const int64_t* positions, int B, int N, int H, int D, float base, float fwd,
                          void* stream) {
    // argument checks mirror curope.cpp:54-59 / kernels.cu:91-94
    A3R_CHECK_ARG(tokens && positions, "a3r_rope2d: null pointer");
    A3R_CHECK_ARG(B > 0 && N > 0 && H > 0, "a3r_rope2d: tokens must have 4 dimensions with positive sizes");
    A3R_CHECK_ARG(D > 0 && D % 4 == 0, "a3r_rope2d: token dim must be multiple of 4 (got %d)", D);
    const long total = (long)B * N * H * (D / 2);
    hipLaunchKernelGGL(rope2d_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), tokens, positions, B, N, H, D,
                       base, fwd);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_rope_table_host(float* cos_host, float* sin_host, int max_pos, float base) {
    A3R_CHECK_ARG(cos_host && sin_host && max_pos > 0, "a3r_rope_table_host: bad argument");
    // RoPE2D.get_cos_sin (pos_embed.py:118-128) with D = 32: inv_freq = 1/(base**(arange(0,32,2)/32)), fp32 steps
    for (int q = 0; q < 16; q++) {
        const float expo = (float)(2 * q) / 32.0f;
        const float inv_freq = 1.0f / powf(base, expo);
        for (int p = 0; p < max_pos; p++) {
            const float fr = (float)p * inv_freq;
            cos_host[p * 16 + q] = cosf(fr);
            sin_host[p * 16 + q] = sinf(fr);
        }
    }
    return A3R_OK;
}

extern "C" int a3r_patchify(const float* img, float* cols, int B, int C, int H, int W, long sb, long sc, long sy, long sx,
                            void* stream) {
    A3R_CHECK_ARG(img && cols, "a3r_patchify: null pointer");
    // patch_embed.py:22-23
    A3R_CHECK_ARG(H > 0 && H % 16 == 0, "Input image height (%d) is not a multiple of patch size (16).", H);
    A3R_CHECK_ARG(W > 0 && W % 16 == 0, "Input image width (%d) is not a multiple of patch size (16).", W);
    const long total = (long)B * (H / 16) * (W / 16) * C * 256;
    ProfScope prof(PK_ELEMENTWISE, 8.0 * total, as_stream(stream));
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), img, cols, B, C, H, W, sb, sc,
                       sy, sx);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

static inline dim3 upsample_grid(int B, int Hc, int Wc, int groups) {      // groups: threads per output pixel
    const long rows = (long)B * Hc;
    return dim3((unsigned)(((long)Wc * groups + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535));
}

extern "C" int a3r_upsample2x(const float* x, float* y, int B, int H, int W, int C, int Hc, int Wc, void* stream) {
    A3R_CHECK_ARG(x && y, "a3r_upsample2x: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "a3r_upsample2x: bad shape");
    A3R_CHECK_ARG(Hc > 0 && Hc <= 2 * H && Wc > 0 && Wc <= 2 * W, "a3r_upsample2x: crop window larger than the 2x map");
    const long total = (long)B * Hc * Wc * (C / 4);
    ProfScope prof(PK_ELEMENTWISE, 16.0 * total + 4.0 * B * H * W * C, as_stream(stream));
    hipLaunchKernelGGL(upsample2x_kernel<0>, upsample_grid(B, Hc, Wc, C / 4), dim3(256), 0, as_stream(stream), x, y, B, H, W, C / 4, Hc, Wc, 1.f, nullptr);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_upsample2x_bf3(const float* x, void* y3, int B, int H, int W, int C, int Hc, int Wc, void* stream) {
    A3R_CHECK_ARG(x && y3, "a3r_upsample2x_bf3: null pointer");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "a3r_upsample2x_bf3: bad shape (C must be a multiple of 8)");
    A3R_CHECK_ARG(Hc > 0 && Hc <= 2 * H && Wc > 0 && Wc <= 2 * W, "a3r_upsample2x_bf3: crop window larger than the 2x map");
    const long total = (long)B * Hc * Wc * (C / 4);
    ProfScope prof(PK_ELEMENTWISE, 24.0 * total + 4.0 * B * H * W * C, as_stream(stream));
    hipLaunchKernelGGL(upsample2x_kernel<1>, upsample_grid(B, Hc, Wc, C / 4), dim3(256), 0, as_stream(stream), x, static_cast<float*>(y3), B, H, W,
                       C / 4, Hc, Wc, 1.f, nullptr);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_upsample2x_fh2(const float* x, void* y2, int B, int H, int W, int C, int Hc, int Wc, float scale, unsigned* absmax,
                                  void* stream) {
    A3R_CHECK_ARG(x && y2, "a3r_upsample2x_fh2: null pointer");
    A3R_CHECK_ARG(scale > 0.f && std::isfinite(scale), "a3r_upsample2x_fh2: scale must be positive and finite");
    A3R_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "a3r_upsample2x_fh2: bad shape (C must be a multiple of 8)");
    A3R_CHECK_ARG(Hc > 0 && Hc <= 2 * H && Wc > 0 && Wc <= 2 * W, "a3r_upsample2x_fh2: crop window larger than the 2x map");
    const long total = (long)B * Hc * Wc * (C / 4);
    ProfScope prof(PK_ELEMENTWISE, 16.0 * total + 4.0 * B * H * W * C, as_stream(stream));
    dim3 grid = upsample_grid(B, Hc, Wc, C / 8);
    // every workgroup loops over ~8 rows and publishes ONE statistics atomic (a long per-thread loop of dependent row loads is
    // latency-bound: 63 rows per workgroup cost +15 % on the full-resolution map; one row per workgroup is half a million atomics)
    const unsigned ycap = (grid.y + 7) / 8;
    if (grid.y > ycap) grid.y = ycap;
    hipLaunchKernelGGL(upsample2x_kernel<2>, grid, dim3(256), 0, as_stream(stream), x, static_cast<float*>(y2), B, H, W,
                       C / 4, Hc, Wc, scale, absmax);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_head_final(const float* x, const float* w, const float* b, float* pts3d, float* conf, long P, int C,
                              void* stream) {
    A3R_CHECK_ARG(x && w && b && pts3d && conf, "a3r_head_final: null pointer");
    A3R_CHECK_ARG(P > 0 && C > 0 && C % 4 == 0, "a3r_head_final: bad shape");
    long blocks = (P + 7) / 8;
    if (blocks > 16384) blocks = 16384;
    ProfScope prof(PK_ELEMENTWISE, 4.0 * P * (C + 4), as_stream(stream));
    if (C == 128 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        long nb = (P + 31) / 32;                                   // a block of four waves takes 32 pixels per step
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(head_final128_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), x, w, b, pts3d, conf, P);
    } else {
        hipLaunchKernelGGL(head_final_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, w, b, pts3d, conf, P, C);
    }
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_pack_conv3x3(const float* w, float* wp, int Cout, int Cin, void* stream) {
    A3R_CHECK_ARG(w && wp && Cout > 0 && Cin > 0, "a3r_pack_conv3x3: bad argument");
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for((long)Cout * Cin * 9)), dim3(256), 0, as_stream(stream), w, wp, Cout, Cin);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}

extern "C" int a3r_pack_convT(const float* w, float* wp, int Cin, int Cout, int s, void* stream) {
    A3R_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && s > 0, "a3r_pack_convT: bad argument");
    hipLaunchKernelGGL(pack_convT_kernel, dim3(grid_for((long)Cin * Cout * s * s)), dim3(256), 0, as_stream(stream), w, wp, Cin, Cout, s);
    A3R_LAUNCH_CHECK();
    return A3R_OK;
}
