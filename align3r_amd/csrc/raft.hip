// RAFT2 ("SEA-RAFT") optical flow -- the flow provider of cloud_opt_flow -- as a launch plan over liba3r.
//
// The reference computes the flow of every pair, both directions, inside the aligner's constructor
// (dust3r/cloud_opt_flow/optimizer.py:118-154) with third_party/RAFT (third_party/raft.py:39-73 builds RAFT2 from
// core/configs/congif_spring_M.json).  Mirrored here, all under /root/reference/third_party/RAFT/core:
//   RAFT2.forward / upsample_data                      raft.py:152-246
//   ResNetFPN (context and feature encoders)           extractor.py:262-350, layer.py:113-141 (BasicBlock)
//   CorrBlock2 (pyramid over down-sampled fmap2)       corr.py:10-60, utils/utils.py:76-91 (bilinear_sampler)
//   BasicUpdateBlock2 / BasicMotionEncoder2            update.py:99-174
//   ConvNextBlock / LayerNorm                          layer.py:35-104
// Round 3: the default arithmetic is the two-plane fp16 form (fh2 kernels, a3r_raft_set_arith(1)); the three-plane bf16 form below
// stays as the fp32-range fallback (a3r_raft_set_arith(0)) that RaftEngine re-runs when a3r_raft_range reports a value outside fp16.
// Every 3x3 convolution and every 1x1 / Linear runs on the exact three-plane bf16 matrix-core kernels (gemm_bf3.hip: fp32-accurate, fp32
// range -- the flow network is a few per cent of a clip's work, so it takes the form that needs no range control); the all-pairs
// correlation is that GEMM with the second feature map as the weight operand.  What has no GEMM shape is here: the 7x7 stems on 3 / 6 /
// 2 channels, the depthwise 7x7 of ConvNeXt, the strided gather of the 1x1 shortcuts, 2x2 averaging, the correlation lookup and the
// convex up-sampling.  BatchNorm (evaluation mode) is folded into the convolutions by the caller (align3r_amd/raft_weights.py
// fold_batchnorm), as are the ConvNeXt layer scale (into pwconv2) and the 0.25 of the up-sampling weights (raft.py:216).
// Maps are channels-last fp32 [B, h, w, C]; all buffers live in the caller's workspace.
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace a3r {

// ------------------------------------------------------------------------------------------- kernels
// direct convolution, k x k, for a handful of input channels (3, 6 or 2): y [B, Ho, Wo, Cout] channels-last.
// Input element (b, c, y, x) of source s sits at xs[s] + b sb + c sc + y sy + x sx and enters as x * in_mul + in_add (the image
// normalisation 2 (x / 255) - 1 of raft.py:194-195); the two sources are concatenated along channels (raft.py:207).
struct DirectConvArgs {
    const float* x0; const float* x1; int c0, c1;
    long sb, sc, sy, sx;
    int B, H, W, k, stride, pad, Ho, Wo, Cout;
    float in_mul, in_add;
    const float* w;      // [(c0 + c1) k k][Cout]  (transposed at finalize)
    const float* bias;
    int relu;
    float* y;
};
// A thread owns PX = 4 consecutive output columns of one output channel; lanes run over the output channels, so the weight loads
// (w transposed to [Cin k k][Cout] at finalize) are coalesced and the input loads are wave-wide broadcasts.  (The first version -- one
// output per thread, weights [Cout][Cin][k][k] -- spent 38 % of the whole flow network in this kernel.)
constexpr int DC_PX = 4;
__global__ __launch_bounds__(256) void direct_conv_kernel(DirectConvArgs a) {
    const int WoG = (a.Wo + DC_PX - 1) / DC_PX;
    const long total = (long)a.B * a.Ho * WoG * a.Cout;
    const int Cin = a.c0 + a.c1;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % a.Cout);
        long p = i / a.Cout;
        const int ox0 = (int)(p % WoG) * DC_PX; p /= WoG;
        const int oy = (int)(p % a.Ho);
        const int b = (int)(p / a.Ho);
        float acc[DC_PX];
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int q = 0; q < DC_PX; q++) acc[q] = bias;
        for (int c = 0; c < Cin; c++) {
            const float* src = (c < a.c0 ? a.x0 + (long)c * a.sc : a.x1 + (long)(c - a.c0) * a.sc) + (long)b * a.sb;
            for (int ky = 0; ky < a.k; ky++) {
                const int iy = oy * a.stride - a.pad + ky;
                if (iy < 0 || iy >= a.H) continue;
                const float* row = src + (long)iy * a.sy;
                for (int kx = 0; kx < a.k; kx++) {
                    const float wv = a.w[(size_t)((c * a.k + ky) * a.k + kx) * a.Cout + co];
#pragma unroll
                    for (int q = 0; q < DC_PX; q++) {
                        const int ix = (ox0 + q) * a.stride - a.pad + kx;
                        if (ix >= 0 && ix < a.W) acc[q] = __fmaf_rn(row[(long)ix * a.sx] * a.in_mul + a.in_add, wv, acc[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < DC_PX; q++)
            if (ox0 + q < a.Wo) a.y[(((long)b * a.Ho + oy) * a.Wo + ox0 + q) * a.Cout + co] = a.relu ? fmaxf(acc[q], 0.f) : acc[q];
    }
}

// The same convolution with the filter size, stride and pixels per thread fixed at compile time (the 7x7 stems, stride 2, and the
// update block's convf1, stride 1): per (channel, filter row) a thread loads the (PX - 1) S + K input columns its PX outputs share
// ONCE (normalised on the way in, zeros outside the image -- fma(0, w, acc) = acc, so the sums are bitwise those of the generic
// kernel, which skips the taps) and then runs K x PX fused multiply-adds from registers; the generic form issued one bounds-checked
// load per multiply-add.  Lanes are consecutive output channels: the input loads are broadcasts, the weight loads coalesced.
template <int K, int S, int PX>
__global__ __launch_bounds__(256) void direct_conv_fixed_kernel(DirectConvArgs a) {
    constexpr int NI = (PX - 1) * S + K;
    const int WoG = (a.Wo + PX - 1) / PX;
    const long total = (long)a.B * a.Ho * WoG * a.Cout;
    const int Cin = a.c0 + a.c1;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % a.Cout);
        long p = i / a.Cout;
        const int ox0 = (int)(p % WoG) * PX; p /= WoG;
        const int oy = (int)(p % a.Ho);
        const int b = (int)(p / a.Ho);
        float acc[PX];
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int q = 0; q < PX; q++) acc[q] = bias;
        const int ix0 = ox0 * S - a.pad;
        for (int c = 0; c < Cin; c++) {
            const float* src = (c < a.c0 ? a.x0 + (long)c * a.sc : a.x1 + (long)(c - a.c0) * a.sc) + (long)b * a.sb;
#pragma unroll 1
            for (int ky = 0; ky < K; ky++) {
                const int iy = oy * S - a.pad + ky;
                if (iy < 0 || iy >= a.H) continue;
                const float* row = src + (long)iy * a.sy;
                float in[NI];
#pragma unroll
                for (int t = 0; t < NI; t++) {
                    const int ix = ix0 + t;
                    in[t] = (ix >= 0 && ix < a.W) ? row[(long)ix * a.sx] * a.in_mul + a.in_add : 0.f;
                }
                const float* wr = a.w + (size_t)((c * K + ky) * K) * a.Cout + co;
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const float wv = wr[(size_t)kx * a.Cout];
#pragma unroll
                    for (int q = 0; q < PX; q++) acc[q] = __fmaf_rn(in[q * S + kx], wv, acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PX; q++)
            if (ox0 + q < a.Wo) a.y[(((long)b * a.Ho + oy) * a.Wo + ox0 + q) * a.Cout + co] = a.relu ? fmaxf(acc[q], 0.f) : acc[q];
    }
}

// depthwise 7x7, padding 3, channels-last x [B, h, w, C]; weights wt [49][C] (transposed at finalize), bias [C].  A thread owns 4
// consecutive output columns of one channel: 7 x 10 input loads for 4 outputs instead of 4 x 49.
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                      float* __restrict__ y, int B, int h, int w, int C) {
    const int wG = (w + 3) / 4;
    const long total = (long)B * h * wG * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long p = i / C;
        const int ox0 = (int)(p % wG) * 4; p /= wG;
        const int oy = (int)(p % h);
        const int b = (int)(p / h);
        float acc[4] = {bias[c], bias[c], bias[c], bias[c]};
        for (int ky = 0; ky < 7; ky++) {
            const int iy = oy - 3 + ky;
            if (iy < 0 || iy >= h) continue;
            const float* row = x + ((long)b * h + iy) * w * C + c;
            float v[10];
#pragma unroll
            for (int j = 0; j < 10; j++) {
                const int ix = ox0 - 3 + j;
                v[j] = (ix >= 0 && ix < w) ? row[(long)ix * C] : 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 7; kx++) {
                const float wv = wt[(ky * 7 + kx) * C + c];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ix = ox0 + q - 3 + kx;
                    if (ix >= 0 && ix < w) acc[q] = __fmaf_rn(v[q + kx], wv, acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (ox0 + q < w) y[(((long)b * h + oy) * w + ox0 + q) * C + c] = acc[q];
    }
}

// y [B, ho, wo, C] = x [B, h, w, C] at (2 oy, 2 ox): the sampling of a 1x1 convolution with stride 2 (layer.py:125)
__global__ __launch_bounds__(256) void gather_s2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h, int w, int ho, int wo, int C4) {
    const long total = (long)B * ho * wo * C4;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x);
    f32x4* yv = reinterpret_cast<f32x4*>(y);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int b = (int)(p / ho);
        yv[i] = xv[(((long)b * h + 2 * oy) * w + 2 * ox) * C4 + c];
    }
}

// F.interpolate(scale_factor=0.5, mode='bilinear', align_corners=False) on channels-last maps (corr.py:22): the source position of
// output o is 2 o + 0.5, i.e. the plain mean of the 2x2 block (0.5 (0.5 a + 0.5 b) + 0.5 (0.5 c + 0.5 d) as ATen evaluates it)
__global__ __launch_bounds__(256) void halve_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h, int w, int ho, int wo, int C) {
    const long total = (long)B * ho * wo * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long p = i / C;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int b = (int)(p / ho);
        const float* r0 = x + (((long)b * h + 2 * oy) * w + 2 * ox) * C + c;
        const float* r1 = r0 + (long)w * C;
        const float t0 = 0.5f * r0[0] + 0.5f * r0[C], t1 = 0.5f * r1[0] + 0.5f * r1[C];
        y[i] = 0.5f * t0 + 0.5f * t1;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, float s, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= s;
}

// dst[r, c0 + c] = src[r, c] (c < Cs): channel concatenation of channels-last maps (torch.cat(dim=1) of the reference's NCHW tensors)
__global__ __launch_bounds__(256) void pack_cols_kernel(float* __restrict__ dst, int ldd, int c0, const float* __restrict__ src, int lds, int Cs, long rows) {
    const long total = rows * Cs;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / Cs;
        const int c = (int)(i - r * Cs);
        dst[r * ldd + c0 + c] = src[r * lds + c];
    }
}

// flow [rows, 2] += upd [rows, ldu][:, 0:2]      (raft.py:233)
__global__ __launch_bounds__(256) void flow_add_kernel(float* __restrict__ flow, const float* __restrict__ upd, int ldu, long rows) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < rows * 2; i += (long)gridDim.x * 256) flow[i] += upd[(i >> 1) * ldu + (i & 1)];
}

// Correlation lookup (corr.py:25-50): for query pixel q = (b, y, x) with target position (x + fx, y + fy), level l, window entry
// (a, c) in [0, 2r]^2: bilinear sample of corr_l[q] ([h_l, w_l]) at (cx / 2^l + (a - r), cy / 2^l + (c - r)) -- the reference stacks
// meshgrid(dy, dx), so the FIRST window index moves x -- with grid_sample's align_corners=True arithmetic (utils.py:79-84: the pixel
// coordinate goes through 2 v / (n - 1) - 1 and back) and zeros outside.  out [B h w, ldo]: channel l (2r+1)^2 + a (2r+1) + c; the
// columns from levels (2r+1)^2 up to ldo are zero-filled (K of the consumer GEMM is padded to a multiple of 32).
struct LookupArgs {
    const float* corr[4]; int hl[4], wl[4];
    const float* flow;        // [B h w, 2]
    float* out; int ldo;
    int B, h, w, r, levels;
};
__global__ __launch_bounds__(256) void corr_lookup_kernel(LookupArgs a) {
    const int win = 2 * a.r + 1, per = win * win, nch = a.levels * per;
    const long total = (long)a.B * a.h * a.w * a.ldo;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % a.ldo);
        const long q = i / a.ldo;
        if (ch >= nch) { a.out[i] = 0.f; continue; }
        const int l = ch / per, e = ch - l * per, wa = e / win, wc = e - wa * win;
        const int qx = (int)(q % a.w), qy = (int)((q / a.w) % a.h);
        const float cx = (float)qx + a.flow[q * 2], cy = (float)qy + a.flow[q * 2 + 1];          // coords_grid2 + flow (raft.py:227)
        const float div = (float)(1 << l);
        const float px = cx / div + (float)(wa - a.r), py = cy / div + (float)(wc - a.r);
        const int H = a.hl[l], W = a.wl[l];
        // bilinear_sampler: normalise, then grid_sample(align_corners=True) un-normalises
        const float gx = 2.f * px / (float)(W - 1) - 1.f, gy = 2.f * py / (float)(H - 1) - 1.f;
        const float ix = (gx + 1.f) * 0.5f * (float)(W - 1), iy = (gy + 1.f) * 0.5f * (float)(H - 1);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float* c = a.corr[l] + q * ((long)H * W);
        auto at = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? c[yy * W + xx] : 0.f; };
        a.out[i] = at(y0, x0) * (wx0 * wy0) + at(y0, x0 + 1) * (wx1 * wy0) + at(y0 + 1, x0) * (wx0 * wy1) + at(y0 + 1, x0 + 1) * (wx1 * wy1);
    }
}

// Convex up-sampling (raft.py:183-199): mask [B h w, 576] with channel k 64 + i 8 + j (k = ky 3 + kx over the 3x3 neighbourhood,
// (i, j) the position inside the 8x8 cell); softmax over k; out[b, c, 8 y + i, 8 x + j] = sum_k p_k 8 flow[b, y + ky - 1, x + kx - 1, c]
// (F.unfold pads with zeros).  out NCHW [B, 2, 8h, 8w].
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask, float* __restrict__ out,
                                                              int B, int h, int w) {
    const long total = (long)B * h * w * 64;
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int ij = (int)(t & 63), i = ij >> 3, j = ij & 7;
        const long q = t >> 6;
        const int x = (int)(q % w), y = (int)((q / w) % h), b = (int)(q / ((long)w * h));
        const float* m = mask + q * 576 + ij;
        float mv[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; k++) { mv[k] = m[k * 64]; mx = fmaxf(mx, mv[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) { mv[k] = expf(mv[k] - mx); den += mv[k]; }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            const float p = mv[k] / den;
            const float* f = flow + (((long)b * h + yy) * w + xx) * 2;
            o0 += p * (8.f * f[0]);
            o1 += p * (8.f * f[1]);
        }
        const long H8 = 8L * h, W8 = 8L * w;
        const long pix = ((long)(8 * y + i)) * W8 + (8 * x + j);
        out[((long)b * 2 + 0) * H8 * W8 + pix] = o0;
        out[((long)b * 2 + 1) * H8 * W8 + pix] = o1;
    }
}

// [N][K] -> [K][N]  (depthwise weights [C][49] -> [49][C]; direct-conv weights [Cout][Cin k k] -> [Cin k k][Cout])
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * K) wt[(size_t)(i % K) * N + i / K] = w[i];
}

static inline unsigned grid1d(long n) {
    const long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 65536 ? 65536 : b);
}

// 7x7 convolutions of 2..6 input channels (stems, convf1): the fixed-size kernel where it applies, the generic one otherwise
static void launch_direct_conv(const DirectConvArgs& a, hipStream_t st) {
    constexpr int PX = 8;
    const long work = (long)a.B * a.Ho * ((a.Wo + PX - 1) / PX) * a.Cout;
    static const bool generic = getenv("A3R_RAFT_DIRECT_CONV") && std::string(getenv("A3R_RAFT_DIRECT_CONV")) == "generic";     // A/B switch
    if (!generic && a.k == 7 && a.stride == 2) hipLaunchKernelGGL((direct_conv_fixed_kernel<7, 2, PX>), dim3(grid1d(work)), dim3(256), 0, st, a);
    else if (!generic && a.k == 7 && a.stride == 1) hipLaunchKernelGGL((direct_conv_fixed_kernel<7, 1, PX>), dim3(grid1d(work)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(direct_conv_kernel, dim3(grid1d((long)a.B * a.Ho * ((a.Wo + DC_PX - 1) / DC_PX) * a.Cout)), dim3(256), 0, st, a);
}

struct RWRef { const float* p = nullptr; std::vector<int64_t> shape; };

}  // namespace a3r
using namespace a3r;

struct a3r_raft_s {
    a3r_raft_config cfg;
    std::map<std::string, RWRef> w;
    bool finalized = false;
    std::map<std::string, const void*> twin;        // conv / linear weight name -> bf3 twin in the packed buffer (weight layout)
    std::map<std::string, const float*> aux;        // depthwise weights transposed
    // the same weights in fh2 form (two fp16 planes of scale * w) with their power-of-two scales: the default arithmetic (round 3)
    std::map<std::string, std::pair<const void*, float>> twin2;
    bool use_fh2 = true;                            // a3r_raft_set_arith
    unsigned* stat = nullptr;                       // device word: max |value written in fh2 form| of the last forward (range check)
};

// ------------------------------------------------------------------------------------------- weight plan
namespace {
struct RItem { std::string name; int kind; int N, K; size_t off, tmp; size_t off2 = 0; };      // kind 0: conv3x3 [N, K/9 ch]; 1: linear [N, K]; 2: depthwise [C=N]; 3: direct conv [N, K = Cin 49], transposed

void resnet_convs(const std::string& p, const a3r_raft_config& c, std::vector<RItem>* v, int output_dim, size_t* off) {
    auto add = [&](const std::string& n, int kind, int N, int K) {
        RItem it{n, kind, N, K, *off, 0, 0};
        *off = align_up(*off + a3r_bf3_w_bytes(N, K), 256);
        if (kind == 0) { it.tmp = *off; *off = align_up(*off + (size_t)N * K * 4, 256); }      // fp32 [Cout, 3, 3, Cin] staging
        v->push_back(it);
    };
    int in_planes = c.initial_dim;
    for (int li = 0; li < 3; li++) {
        const int dim = c.block_dims[li];
        for (int bi = 0; bi < c.n_blocks[li]; bi++) {
            const std::string q = p + ".layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            const int cin = bi == 0 ? in_planes : dim, stride = bi == 0 && li > 0 ? 2 : 1;
            add(q + ".conv1.weight", 0, dim, 9 * cin);
            add(q + ".conv2.weight", 0, dim, 9 * dim);
            if (!(stride == 1 && cin == dim)) add(q + ".downsample.0.weight", 1, dim, cin);
        }
        in_planes = dim;
    }
    add(p + ".final_conv.weight", 1, output_dim, c.block_dims[2]);
}

std::vector<RItem> raft_pack_plan(const a3r_raft_config& c, size_t* total) {
    std::vector<RItem> v;
    size_t off = 0;
    const int d = c.dim;
    auto add = [&](const std::string& n, int kind, int N, int K) {
        RItem it{n, kind, N, K, off, 0, 0};
        if (kind == 2) { off = align_up(off + (size_t)N * 49 * 4, 256); v.push_back(it); return; }
        if (kind == 3) { off = align_up(off + (size_t)N * K * 4, 256); v.push_back(it); return; }      // direct-conv weight, transposed
        off = align_up(off + a3r_bf3_w_bytes(N, K), 256);
        if (kind == 0) { it.tmp = off; off = align_up(off + (size_t)N * K * 4, 256); }
        v.push_back(it);
    };
    add("cnet.conv1.weight", 3, c.initial_dim, 6 * 49);
    add("fnet.conv1.weight", 3, c.initial_dim, 3 * 49);
    add("update_block.encoder.convf1.weight", 3, d, 2 * 49);
    resnet_convs("cnet", c, &v, 2 * d, &off);
    resnet_convs("fnet", c, &v, 2 * d, &off);
    add("init_conv.weight", 0, 2 * d, 9 * 2 * d);
    add("upsample_weight.0.weight", 0, 2 * d, 9 * d);
    add("upsample_weight.2.weight", 1, 576, 2 * d);
    add("flow_head.0.weight", 0, 2 * d, 9 * d);
    add("flow_head.2.weight", 0, 6, 9 * 2 * d);
    const int cc = c.corr_levels * (2 * c.radius + 1) * (2 * c.radius + 1), ccp = (cc + 31) / 32 * 32;
    const std::string e = "update_block.encoder.";
    add(e + "convc1.weight", 1, 2 * d, ccp);
    add(e + "convc2.weight", 0, d + d / 2, 9 * 2 * d);
    add(e + "convf2.weight", 0, d / 2, 9 * d);
    add(e + "conv.weight", 0, d - 2, 9 * 2 * d);
    for (int i = 0; i < c.num_blocks; i++) {
        const std::string q = "update_block.refine." + std::to_string(i) + ".";
        add(q + "dwconv.weight", 2, 3 * d, 49);
        add(q + "pwconv1.weight", 1, 4 * d, 3 * d);
        add(q + "pwconv2.weight", 1, 3 * d, 4 * d);
        add(q + "final.weight", 1, d, 3 * d);
    }
    // second image of every conv / linear weight: fh2 form (4 bytes per element); then one line for the range statistic and the
    // absmax scratch word of the packing pass
    for (RItem& it : v)
        if (it.kind == 0 || it.kind == 1) { it.off2 = off; off = align_up(off + (size_t)it.N * it.K * 4, 256); }
    off = align_up(off + 256, 256);
    *total = off;
    return v;
}

int rneed(a3r_raft_s* m, const std::string& name, std::vector<int64_t> shape, const float** out) {
    auto it = m->w.find(name);
    if (it == m->w.end()) { set_error("a3r_raft_finalize: missing weight '%s'", name.c_str()); return A3R_ESTATE; }
    if (it->second.shape != shape) {
        std::string got, want;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        for (auto d : shape) want += std::to_string(d) + ",";
        set_error("a3r_raft_finalize: weight '%s' has shape [%s] but [%s] is required", name.c_str(), got.c_str(), want.c_str());
        return A3R_EINVAL;
    }
    *out = it->second.p;
    return A3R_OK;
}

// fh2 image of one weight ([N, K] fp32 at src) with its power-of-two scale (max |w| into [2^12, 2^13))
int pack_fh2(a3r_raft_s* m, const RItem& it, const float* src, char* pk, size_t total, void* stream) {
    float* scratch = reinterpret_cast<float*>(pk + total - 128);
    if (int rc = a3r_absmax(src, (long)it.N * it.K, scratch, stream)) return rc;
    float amax = 0.f;
    if (hipMemcpyAsync(&amax, scratch, 4, hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess || hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
        set_error("a3r_raft_finalize: reading the weight range failed");
        return A3R_EHIP;
    }
    A3R_CHECK_ARG(std::isfinite(amax), "a3r_raft_finalize: weight '%s' is not finite", it.name.c_str());
    const float scale = a3r_fh2_weight_scale(amax);
    if (int rc = a3r_split_fh2(src, it.K, pk + it.off2, it.N, it.K, scale, nullptr, stream)) return rc;
    m->twin2[it.name] = {pk + it.off2, scale};
    return A3R_OK;
}
}  // namespace

extern "C" int a3r_raft_create(const a3r_raft_config* cfg, a3r_raft_t* out) {
    A3R_CHECK_ARG(cfg && out, "a3r_raft_create: null argument");
    A3R_CHECK_ARG(cfg->initial_dim % 32 == 0 && cfg->dim % 64 == 0, "a3r_raft_create: initial_dim must be a multiple of 32 and dim of 64");
    for (int i = 0; i < 3; i++)
        A3R_CHECK_ARG(cfg->block_dims[i] % 32 == 0 && cfg->n_blocks[i] >= 1, "a3r_raft_create: block_dims must be multiples of 32");
    A3R_CHECK_ARG(cfg->corr_levels >= 1 && cfg->corr_levels <= 4 && cfg->radius >= 1 && cfg->radius <= 8 && cfg->num_blocks >= 1,
                  "a3r_raft_create: corr_levels in 1..4, radius in 1..8, num_blocks >= 1");
    a3r_raft_s* m = new (std::nothrow) a3r_raft_s();
    A3R_CHECK_ARG(m, "out of host memory");
    m->cfg = *cfg;
    *out = m;
    return A3R_OK;
}

extern "C" int a3r_raft_destroy(a3r_raft_t m) {
    delete m;
    return A3R_OK;
}

extern "C" int a3r_raft_set_weight(a3r_raft_t m, const char* name, const float* ptr, int ndim, const int64_t* shape) {
    A3R_CHECK_ARG(m && name && ptr && ndim >= 1 && ndim <= 4 && shape, "a3r_raft_set_weight: bad argument");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "a3r_raft_set_weight: %s is not 16-byte aligned", name);
    RWRef r;
    r.p = ptr;
    r.shape.assign(shape, shape + ndim);
    m->w[name] = r;
    m->finalized = false;
    return A3R_OK;
}

extern "C" size_t a3r_raft_packed_bytes(a3r_raft_t m) {
    if (!m) return 0;
    size_t total = 0;
    raft_pack_plan(m->cfg, &total);
    return total;
}

extern "C" int a3r_raft_finalize(a3r_raft_t m, void* packed, size_t packed_bytes, void* stream) {
    A3R_CHECK_ARG(m && packed, "a3r_raft_finalize: null argument");
    size_t total = 0;
    std::vector<RItem> plan = raft_pack_plan(m->cfg, &total);
    A3R_CHECK_ARG(packed_bytes >= total, "a3r_raft_finalize: packed buffer too small (%zu < %zu)", packed_bytes, total);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 255) == 0, "a3r_raft_finalize: packed buffer must be 256-byte aligned");
    char* pk = static_cast<char*>(packed);
    m->twin.clear();
    m->twin2.clear();
    m->aux.clear();
    m->stat = reinterpret_cast<unsigned*>(pk + total - 256);
    for (const RItem& it : plan) {
        const float* src;
        if (it.kind == 0) {
            const int cin = it.K / 9;
            if (int rc = rneed(m, it.name, {it.N, cin, 3, 3}, &src)) return rc;
            float* tmp = reinterpret_cast<float*>(pk + it.tmp);
            if (int rc = a3r_pack_conv3x3(src, tmp, it.N, cin, stream)) return rc;
            if (int rc = a3r_split_bf3_w(tmp, it.K, pk + it.off, it.N, it.K, stream)) return rc;
            m->twin[it.name] = pk + it.off;
            if (int rc = pack_fh2(m, it, tmp, pk, total, stream)) return rc;
        } else if (it.kind == 1) {
            // Linear weights are [N, K]; 1x1 convolutions [N, K, 1, 1]
            auto wi = m->w.find(it.name);
            if (wi == m->w.end()) { set_error("a3r_raft_finalize: missing weight '%s'", it.name.c_str()); return A3R_ESTATE; }
            const std::vector<int64_t>& sh = wi->second.shape;
            const bool ok = (sh.size() == 2 && sh[0] == it.N && sh[1] == it.K) || (sh.size() == 4 && sh[0] == it.N && sh[1] == it.K && sh[2] == 1 && sh[3] == 1);
            A3R_CHECK_ARG(ok, "a3r_raft_finalize: weight '%s' must be [%d, %d] (or [%d, %d, 1, 1])", it.name.c_str(), it.N, it.K, it.N, it.K);
            if (int rc = a3r_split_bf3_w(wi->second.p, it.K, pk + it.off, it.N, it.K, stream)) return rc;
            m->twin[it.name] = pk + it.off;
            if (int rc = pack_fh2(m, it, wi->second.p, pk, total, stream)) return rc;
        } else if (it.kind == 3) {
            if (int rc = rneed(m, it.name, {it.N, it.K / 49, 7, 7}, &src)) return rc;
            float* wt = reinterpret_cast<float*>(pk + it.off);
            hipLaunchKernelGGL(transpose_kernel, dim3((it.N * it.K + 255) / 256), dim3(256), 0, as_stream(stream), src, wt, it.N, it.K);
            A3R_LAUNCH_CHECK();
            m->aux[it.name] = wt;
        } else {
            if (int rc = rneed(m, it.name, {it.N, 1, 7, 7}, &src)) return rc;
            float* wt = reinterpret_cast<float*>(pk + it.off);
            hipLaunchKernelGGL(transpose_kernel, dim3((it.N * 49 + 255) / 256), dim3(256), 0, as_stream(stream), src, wt, it.N, 49);
            A3R_LAUNCH_CHECK();
            m->aux[it.name] = wt;
        }
    }
    m->finalized = true;
    return A3R_OK;
}

// ------------------------------------------------------------------------------------------- launch plan
namespace {
struct RArena {
    char* base; size_t off, cap; bool dry; size_t peak;
    float* alloc(size_t nfloat) {
        const size_t o = off;
        off = align_up(off + nfloat * 4, 256);
        if (off > peak) peak = off;
        return dry ? nullptr : reinterpret_cast<float*>(base + o);
    }
    float* alloc3(size_t rows, int K) { return alloc(rows * K * 3 / 2); }      // a bf3 [rows, K] matrix
};

struct RPlan {
    a3r_raft_s* m;
    RArena ar;
    void* stream;
    int rc = A3R_OK;
    bool skip() const { return ar.dry || rc != A3R_OK; }
    const float* wptr(const std::string& n) {
        auto it = m->w.find(n);
        if (it == m->w.end()) { if (!rc) { set_error("a3r_raft_forward: missing weight '%s'", n.c_str()); rc = A3R_ESTATE; } return nullptr; }
        return it->second.p;
    }
    const float* auxw(const std::string& n) {
        auto it = m->aux.find(n);
        if (it == m->aux.end()) { if (!rc) { set_error("a3r_raft_forward: weight '%s' was not packed", n.c_str()); rc = A3R_ESTATE; } return nullptr; }
        return it->second;
    }
    const void* twin(const std::string& n) {
        auto it = m->twin.find(n);
        if (it == m->twin.end()) { if (!rc) { set_error("a3r_raft_forward: weight '%s' was not packed", n.c_str()); rc = A3R_ESTATE; } return nullptr; }
        return it->second;
    }
    a3r_epilogue epi(int kind, const float* bias, const float* resid = nullptr) {
        a3r_epilogue e = {};
        e.epi = kind; e.bias = bias; e.resid = resid;
        return e;
    }
    // The plan below is written for the three-plane bf16 form ("x3" operands, out_bf3 / aux_bf3 epilogues).  With m->use_fh2 the
    // same calls run on the fh2 kernels (3 fp16 passes instead of 6 bf16 ones, 4 instead of 6 operand bytes): operands are fh2
    // matrices with scale 1 (they fit the buffers sized for bf3), out_bf3 / aux_bf3 mean out_fh2 / aux_fh2, and every fh2 producer
    // reports max |stored value| into m->stat, which the caller checks against fp16's range after the forward.
    bool fh2() const { return m->use_fh2; }
    std::pair<const void*, float> twin2(const std::string& n) {
        auto it = m->twin2.find(n);
        if (it == m->twin2.end()) { if (!rc) { set_error("a3r_raft_forward: weight '%s' was not packed", n.c_str()); rc = A3R_ESTATE; } return {nullptr, 1.f}; }
        return it->second;
    }
    void to_fh2(a3r_epilogue& e) {
        e.out_fh2 = e.out_bf3; e.out_bf3 = 0;
        e.aux_fh2 = e.aux_bf3; e.aux_bf3 = nullptr;
        e.out_absmax = (e.out_fh2 || e.aux_fh2) ? m->stat : nullptr;
    }
    // 3x3 convolution `name` (weight / bias) on the bf3 / fh2 map x3 [B, H, W, Cin]
    void conv3(const float* x3, const std::string& name, float* y, int B, int H, int W, int Cin, int Cout, int stride, a3r_epilogue e) {
        if (skip()) return;
        e.bias = wptr(name + ".bias");
        if (fh2()) {
            const auto w2 = twin2(name + ".weight");
            to_fh2(e);
            if (!rc) rc = a3r_conv3x3_fh2(x3, w2.first, w2.second, y, B, H, W, Cin, Cout, stride, &e, stream);
            return;
        }
        const void* w3 = twin(name + ".weight");
        if (!rc) rc = a3r_conv3x3_bf3(x3, w3, y, B, H, W, Cin, Cout, stride, &e, stream);
    }
    void linear(const float* x3, const std::string& name, float* y, int ldc, long M, int N, int K, a3r_epilogue e) {
        if (skip()) return;
        e.bias = wptr(name + ".bias");
        if (fh2()) {
            const auto w2 = twin2(name + ".weight");
            to_fh2(e);
            if (!rc) rc = a3r_linear_fh2(x3, w2.first, w2.second, y, ldc, (int)M, N, K, &e, stream);
            return;
        }
        const void* w3 = twin(name + ".weight");
        if (!rc) rc = a3r_linear_bf3(x3, w3, y, ldc, (int)M, N, K, &e, stream);
    }
    void split(const float* x, int ldx, float* y3, long M, int K) {
        if (skip()) return;
        rc = fh2() ? a3r_split_fh2(x, ldx, y3, M, K, 1.f, m->stat, stream) : a3r_split_bf3(x, ldx, y3, M, K, stream);
    }
    template <class F> void launch(F&& f) {
        if (skip()) return;
        f(as_stream(stream));
        if (hipGetLastError() != hipSuccess) { set_error("a3r_raft_forward: kernel launch failed"); rc = A3R_EHIP; }
    }
    void pack_cols(float* dst, int ldd, int c0, const float* src, int lds, int Cs, long rows) {
        launch([&](hipStream_t st) { hipLaunchKernelGGL(pack_cols_kernel, dim3(grid1d(rows * Cs)), dim3(256), 0, st, dst, ldd, c0, src, lds, Cs, rows); });
    }
};

// ResNetFPN.forward (extractor.py:338-350) after the stem: s = relu(bn1(conv1(x))) [nimg, H2, W2, C0] fp32 is given.  Writes
// final_conv's output to out (fp32 [nimg, h, w, out_dim]) or, when out3 is given, in bf3 form to out3.
void resnet(RPlan& P, const std::string& p, float* s, int nimg, int H2, int W2, float* out, float* out3, int out_dim) {
    const a3r_raft_config& c = P.m->cfg;
    RArena& ar = P.ar;
    int h = H2, w = W2, in_planes = c.initial_dim;
    const float* x = s;                                  // fp32 block input
    float* x3 = ar.alloc3((size_t)nimg * h * w, in_planes);
    P.split(x, in_planes, x3, (long)nimg * h * w, in_planes);
    for (int li = 0; li < 3; li++) {
        const int dim = c.block_dims[li];
        for (int bi = 0; bi < c.n_blocks[li]; bi++) {
            const std::string q = p + ".layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            const int cin = bi == 0 ? in_planes : dim, stride = bi == 0 && li > 0 ? 2 : 1;
            const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
            const size_t px = (size_t)nimg * ho * wo;
            // y = relu(bn1(conv1(x)))                                                        layer.py:134
            float* y3 = ar.alloc3(px, dim);
            a3r_epilogue e1 = P.epi(A3R_EPI_RELU, nullptr);
            e1.out_bf3 = 1;
            P.conv3(x3, q + ".conv1", y3, nimg, h, w, cin, dim, stride, e1);
            // shortcut: x, or bn3(conv1x1 stride s (x))                                      layer.py:122-129,137-138
            const float* sc = x;
            if (!(stride == 1 && cin == dim)) {
                const float* g3 = x3;
                if (stride == 2) {
                    float* gx = ar.alloc(px * cin);
                    P.launch([&](hipStream_t st) {
                        hipLaunchKernelGGL(gather_s2_kernel, dim3(grid1d((long)px * cin / 4)), dim3(256), 0, st, x, gx, nimg, h, w, ho, wo, cin / 4);
                    });
                    float* gx3 = ar.alloc3(px, cin);
                    P.split(gx, cin, gx3, (long)px, cin);
                    g3 = gx3;
                }
                float* scb = ar.alloc(px * dim);
                P.linear(g3, q + ".downsample.0", scb, dim, (long)px, dim, cin, P.epi(A3R_EPI_NONE, nullptr));
                sc = scb;
            }
            // y = relu(bn2(conv2(y))); out = relu(shortcut + y)                              layer.py:135-141: ONE launch
            float* o = ar.alloc(px * dim);
            float* o3 = ar.alloc3(px, dim);
            a3r_epilogue e2 = P.epi(A3R_EPI_RESID, nullptr, sc);
            e2.relu_acc = 1; e2.relu_out = 1; e2.aux_bf3 = o3;
            P.conv3(y3, q + ".conv2", o, nimg, ho, wo, dim, dim, 1, e2);
            x = o; x3 = o3; h = ho; w = wo;
        }
        in_planes = dim;
    }
    a3r_epilogue ef = P.epi(A3R_EPI_NONE, nullptr);
    if (out3) ef.out_bf3 = 1;
    P.linear(x3, p + ".final_conv", out3 ? out3 : out, out_dim, (long)nimg * h * w, out_dim, c.block_dims[2], ef);
}

// RAFT2.forward(test_mode=True) (raft.py:185-246) for B pairs.  img1 / img2 [B, 3, H, W] with values in [0, 255]; flow_out [B, 2, H, W]
// (the last prediction; the reference up-samples every iteration's flow and returns the list, its caller keeps `[1]`).
// taps: optional intermediate copies for the parity tests (any pointer may be null).
// phase 0: the whole forward (fmap1_in / fmap2_in given: the feature network is skipped and those per-frame features are used);
// phase 1: the feature network alone on B frames `img1` -> fmap_out [B, h, w, 2 dim] (a3r_raft_encode)
int raft_plan(a3r_raft_s* m, bool dry, const float* img1, const float* img2, int B, int H, int W, int iters, float* flow_out, void* ws,
              size_t ws_bytes, void* stream, size_t* peak, const a3r_raft_taps* taps, int phase = 0, const float* fmap1_in = nullptr,
              const float* fmap2_in = nullptr, float* fmap_out = nullptr) {
    const a3r_raft_config& c = m->cfg;
    const int d = c.dim, h = H / 8, w = W / 8, H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1;
    const long hw = (long)h * w, Bhw = (long)B * hw;
    const int win = 2 * c.radius + 1, cc = c.corr_levels * win * win, ccp = (cc + 31) / 32 * 32;
    RPlan P;
    P.m = m; P.stream = stream;
    P.ar = {static_cast<char*>(ws), 0, ws_bytes, dry, 0};
    RArena& ar = P.ar;
    if (!dry && m->use_fh2 && m->stat && hipMemsetAsync(m->stat, 0, 4, as_stream(stream)) != hipSuccess) {
        set_error("a3r_raft_forward: clearing the range statistic failed");
        return A3R_EHIP;
    }
    auto tap = [&](float* dst, const float* src, size_t n) {
        if (dry || P.rc || !dst) return;
        if (hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) { set_error("a3r_raft_forward: tap copy failed"); P.rc = A3R_EHIP; }
    };
    if (phase == 1) {
        // the feature network on B frames: fnet(2 x / 255 - 1) (raft.py:222-223); its rows do not depend on what else is in the batch
        float* s = ar.alloc((size_t)B * H2 * W2 * c.initial_dim);
        P.launch([&](hipStream_t st) {
            DirectConvArgs a = {img1, nullptr, 3, 0, 3L * H * W, (long)H * W, W, 1, B, H, W, 7, 2, 3, H2, W2, c.initial_dim, 2.f / 255.f, -1.f,
                                P.auxw("fnet.conv1.weight"), P.wptr("fnet.conv1.bias"), 1, s};
            launch_direct_conv(a, st);
        });
        float* fo = dry ? ar.alloc(Bhw * 2 * d) : fmap_out;
        resnet(P, "fnet", s, B, H2, W2, fo, nullptr, 2 * d);
        if (peak) *peak = ar.peak;
        return P.rc;
    }
    // ---------------- persistent buffers
    float* cn = ar.alloc(Bhw * 2 * d);                  // init_conv output: [net | context]
    float* fm3 = ar.alloc3(2 * Bhw, 2 * d);             // fnet(image1), fnet(image2) in bf3 form (rows: all of image1, then image2)
    float* fm = ar.alloc(2 * Bhw * 2 * d);              // the same in fp32 (the pyramid's 2x2 means, the taps)
    float* corr[4] = {nullptr, nullptr, nullptr, nullptr};
    int hl[4], wl[4];
    {
        int a = h, b = w;
        for (int l = 0; l < c.corr_levels; l++) { hl[l] = a; wl[l] = b; corr[l] = ar.alloc((size_t)Bhw * a * b); a /= 2; b /= 2; }
    }
    float* net = ar.alloc(Bhw * d);
    float* X = ar.alloc(Bhw * 3 * d);                   // [net | context | motion features]: the ConvNeXt blocks' input (update.py:170-173)
    float* flow8 = ar.alloc(Bhw * 2);
    float* fu = ar.alloc(Bhw * 6);                      // flow_head output: flow update (2) + info (4)
    float* wgt = ar.alloc(Bhw * 576);
    const size_t mark = ar.off;
    // ---------------- context network on cat(image1, image2) (raft.py:207-209)
    {
        float* s = ar.alloc((size_t)B * H2 * W2 * c.initial_dim);
        P.launch([&](hipStream_t st) {
            DirectConvArgs a = {img1, img2, 3, 3, 3L * H * W, (long)H * W, W, 1, B, H, W, 7, 2, 3, H2, W2, c.initial_dim, 2.f / 255.f, -1.f,
                                P.auxw("cnet.conv1.weight"), P.wptr("cnet.conv1.bias"), 1, s};
            launch_direct_conv(a, st);
        });
        float* c3 = ar.alloc3(Bhw, 2 * d);
        resnet(P, "cnet", s, B, H2, W2, nullptr, c3, 2 * d);
        P.conv3(c3, "init_conv", cn, B, h, w, 2 * d, 2 * d, 1, P.epi(A3R_EPI_NONE, nullptr));
    }
    ar.off = mark;
    tap(taps ? taps->cnet : nullptr, cn, (size_t)Bhw * 2 * d);
    // net, context = split(cnet) (raft.py:210)
    P.pack_cols(net, d, 0, cn, 2 * d, d, Bhw);
    P.pack_cols(X, 3 * d, d, cn + d, 2 * d, d, Bhw);
    // ---------------- feature network on both images (raft.py:222-223) and the correlation pyramid (corr.py:11-23)
    if (fmap1_in && fmap2_in) {                                      // per-frame features computed once by a3r_raft_encode
        if (!P.skip()) {
            const size_t bytes = (size_t)Bhw * 2 * d * 4;
            if (hipMemcpyAsync(fm, fmap1_in, bytes, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess ||
                hipMemcpyAsync(fm + (size_t)Bhw * 2 * d, fmap2_in, bytes, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) {
                set_error("a3r_raft_forward_features: copy failed"); P.rc = A3R_EHIP;
            }
        }
    } else {
        float* s = ar.alloc((size_t)2 * B * H2 * W2 * c.initial_dim);
        for (int k = 0; k < 2; k++)
            P.launch([&](hipStream_t st) {
                DirectConvArgs a = {k ? img2 : img1, nullptr, 3, 0, 3L * H * W, (long)H * W, W, 1, B, H, W, 7, 2, 3, H2, W2, c.initial_dim, 2.f / 255.f, -1.f,
                                    P.auxw("fnet.conv1.weight"), P.wptr("fnet.conv1.bias"), 1, s + (size_t)k * B * H2 * W2 * c.initial_dim};
                launch_direct_conv(a, st);
            });
        resnet(P, "fnet", s, 2 * B, H2, W2, fm, nullptr, 2 * d);
    }
    ar.off = mark;
    tap(taps ? taps->fmap : nullptr, fm, (size_t)2 * Bhw * 2 * d);
    {
        // corr_l[b] = fmap1[b] @ fmap2_l[b]^T / sqrt(dim) (corr.py:53-60): fmap1 is scaled once, fmap2_l enters as the GEMM's weight operand
        const int D = 2 * d;
        float* f1s = ar.alloc(Bhw * D);
        if (!P.skip()) {
            if (hipMemcpyAsync(f1s, fm, (size_t)Bhw * D * 4, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) { set_error("a3r_raft_forward: copy failed"); P.rc = A3R_EHIP; }
        }
        P.launch([&](hipStream_t st) { hipLaunchKernelGGL(scale_kernel, dim3(grid1d(Bhw * D)), dim3(256), 0, st, f1s, 1.f / sqrtf((float)D), Bhw * D); });
        P.split(f1s, D, fm3, Bhw, D);
        float* f2 = fm + (size_t)Bhw * D;                           // level 0 of fmap2
        float* f2n = ar.alloc((size_t)B * (h / 2) * (w / 2) * D);
        float* f2m = ar.alloc((size_t)B * (h / 4 > 0 ? h / 4 : 1) * (w / 4 > 0 ? w / 4 : 1) * D);
        float* w3 = ar.alloc((size_t)a3r_bf3_w_bytes(hw, D) / 4 + 64);
        for (int l = 0; l < c.corr_levels; l++) {
            const long n2 = (long)hl[l] * wl[l];
            for (int b = 0; b < B; b++) {
                if (P.skip()) break;
                a3r_epilogue e = {};
                if (P.fh2()) {
                    P.rc = a3r_split_fh2(f2 + (size_t)b * n2 * D, D, w3, n2, D, 1.f, m->stat, stream);
                    if (!P.rc) P.rc = a3r_linear_fh2(reinterpret_cast<char*>(fm3) + (size_t)b * hw * D * 4, w3, 1.f, corr[l] + (size_t)b * hw * n2, (int)n2, (int)hw, (int)n2, D, &e, stream);
                    continue;
                }
                P.rc = a3r_split_bf3_w(f2 + (size_t)b * n2 * D, D, w3, n2, D, stream);
                if (!P.rc) P.rc = a3r_linear_bf3(reinterpret_cast<char*>(fm3) + (size_t)b * hw * D * 6, w3, corr[l] + (size_t)b * hw * n2, (int)n2, (int)hw, (int)n2, D, &e, stream);
            }
            if (l + 1 < c.corr_levels) {                             // fmap2 <- interpolate(fmap2, 0.5) (corr.py:22)
                float* dst = (f2 == f2n) ? f2m : f2n;
                const int ha = hl[l], wa = wl[l], hb = hl[l + 1], wb = wl[l + 1];
                const float* src = f2;
                P.launch([&](hipStream_t st) { hipLaunchKernelGGL(halve_kernel, dim3(grid1d((long)B * hb * wb * D)), dim3(256), 0, st, src, dst, B, ha, wa, hb, wb, D); });
                f2 = dst;
            }
        }
    }
    ar.off = mark;
    if (taps) for (int l = 0; l < c.corr_levels; l++) tap(taps->corr_pyr[l], corr[l], (size_t)Bhw * hl[l] * wl[l]);
    // ---------------- heads on a hidden state (raft.py:213-216, 230-231): fu = flow_head(net), wgt = .25 * upsample_weight(net)
    float* n3 = ar.alloc3(Bhw, d);
    float* t3 = ar.alloc3(Bhw, 2 * d);
    auto heads = [&]() {
        P.split(net, d, n3, Bhw, d);
        a3r_epilogue er = P.epi(A3R_EPI_RELU, nullptr);
        er.out_bf3 = 1;
        P.conv3(n3, "flow_head.0", t3, B, h, w, d, 2 * d, 1, er);
        P.conv3(t3, "flow_head.2", fu, B, h, w, 2 * d, 6, 1, P.epi(A3R_EPI_NONE, nullptr));
        P.conv3(n3, "upsample_weight.0", t3, B, h, w, d, 2 * d, 1, er);
        P.linear(t3, "upsample_weight.2", wgt, 576, Bhw, 576, 2 * d, P.epi(A3R_EPI_NONE, nullptr));
    };
    heads();
    tap(taps ? taps->flow_update0 : nullptr, fu, (size_t)Bhw * 6);
    tap(taps ? taps->weight0 : nullptr, wgt, (size_t)Bhw * 576);
    P.pack_cols(flow8, 2, 0, fu, 6, 2, Bhw);                         // flow_8x = flow_update[:, :2] (raft.py:217)
    // ---------------- iterations (raft.py:225-238)
    float* lk = ar.alloc(Bhw * ccp);
    float* lk3 = ar.alloc3(Bhw, ccp);
    float* c13 = ar.alloc3(Bhw, 2 * d);
    float* cf = ar.alloc(Bhw * 2 * d);                                // cat([cor, flo]) (update.py:113)
    float* tmp = ar.alloc(Bhw * 2 * d);
    float* f1 = ar.alloc(Bhw * d);
    float* f13 = ar.alloc3(Bhw, d);
    float* cf3 = ar.alloc3(Bhw, 2 * d);
    float* dw = ar.alloc(Bhw * 3 * d);
    float* ln3 = ar.alloc3(Bhw, 3 * d);
    float* hid3 = ar.alloc3(Bhw, 4 * d);
    float* S = ar.alloc(Bhw * 3 * d);
    float* S3 = ar.alloc3(Bhw, 3 * d);
    const std::string enc = "update_block.encoder.";
    for (int it = 0; it < iters; it++) {
        // corr = corr_fn(coords_grid + flow_8x) (raft.py:227-228)
        P.launch([&](hipStream_t st) {
            LookupArgs a = {};
            for (int l = 0; l < c.corr_levels; l++) { a.corr[l] = corr[l]; a.hl[l] = hl[l]; a.wl[l] = wl[l]; }
            a.flow = flow8; a.out = lk; a.ldo = ccp; a.B = B; a.h = h; a.w = w; a.r = c.radius; a.levels = c.corr_levels;
            hipLaunchKernelGGL(corr_lookup_kernel, dim3(grid1d(Bhw * ccp)), dim3(256), 0, st, a);
        });
        if (it == 0) tap(taps ? taps->lookup0 : nullptr, lk, (size_t)Bhw * ccp);
        // BasicMotionEncoder2 (update.py:99-117)
        P.split(lk, ccp, lk3, Bhw, ccp);
        a3r_epilogue er = P.epi(A3R_EPI_RELU, nullptr);
        er.out_bf3 = 1;
        P.linear(lk3, enc + "convc1", c13, 2 * d, Bhw, 2 * d, ccp, er);                                     // cor = relu(convc1(corr))
        P.conv3(c13, enc + "convc2", tmp, B, h, w, 2 * d, d + d / 2, 1, P.epi(A3R_EPI_RELU, nullptr));     // cor = relu(convc2(cor))
        P.pack_cols(cf, 2 * d, 0, tmp, d + d / 2, d + d / 2, Bhw);
        P.launch([&](hipStream_t st) {                                                                      // flo = relu(convf1(flow)): 7x7 on 2 channels
            DirectConvArgs a = {flow8, nullptr, 2, 0, hw * 2, 1, (long)w * 2, 2, B, h, w, 7, 1, 3, h, w, d, 1.f, 0.f,
                                P.auxw(enc + "convf1.weight"), P.wptr(enc + "convf1.bias"), 1, f1};
            launch_direct_conv(a, st);
        });
        P.split(f1, d, f13, Bhw, d);
        P.conv3(f13, enc + "convf2", tmp, B, h, w, d, d / 2, 1, P.epi(A3R_EPI_RELU, nullptr));              // flo = relu(convf2(flo))
        P.pack_cols(cf, 2 * d, d + d / 2, tmp, d / 2, d / 2, Bhw);
        P.split(cf, 2 * d, cf3, Bhw, 2 * d);
        P.conv3(cf3, enc + "conv", tmp, B, h, w, 2 * d, d - 2, 1, P.epi(A3R_EPI_RELU, nullptr));            // out = relu(conv(cat))
        P.pack_cols(X, 3 * d, 2 * d, tmp, d - 2, d - 2, Bhw);                                                // motion = cat([out, flow])
        P.pack_cols(X, 3 * d, 3 * d - 2, flow8, 2, 2, Bhw);
        if (it == 0 && taps && taps->motion0 && !P.skip()) {
            // (tap: the motion features alone, [Bhw, d])
            hipLaunchKernelGGL(pack_cols_kernel, dim3(grid1d(Bhw * d)), dim3(256), 0, as_stream(stream), taps->motion0, d, 0, X + 2 * d, 3 * d, d, Bhw);
        }
        // refine: net = ConvNextBlock_i(cat([net, inp])) (update.py:170-173, layer.py:35-70)
        for (int r = 0; r < c.num_blocks; r++) {
            const std::string q = "update_block.refine." + std::to_string(r) + ".";
            P.pack_cols(X, 3 * d, 0, net, d, d, Bhw);
            P.launch([&](hipStream_t st) {
                auto wi = P.m->aux.find(q + "dwconv.weight");
                hipLaunchKernelGGL(dwconv7_kernel, dim3(grid1d((long)B * h * ((w + 3) / 4) * 3 * d)), dim3(256), 0, st, X, wi->second, P.wptr(q + "dwconv.bias"), dw, B, h, w, 3 * d);
            });
            if (!P.skip())
                P.rc = P.fh2() ? a3r_layernorm_fh2(dw, P.wptr(q + "norm.weight"), P.wptr(q + "norm.bias"), ln3, (int)Bhw, 3 * d, 1e-6f, 1.f, m->stat, stream)
                               : a3r_layernorm_bf3(dw, P.wptr(q + "norm.weight"), P.wptr(q + "norm.bias"), ln3, (int)Bhw, 3 * d, 1e-6f, 0, stream);
            a3r_epilogue eg = P.epi(A3R_EPI_GELU, nullptr);
            eg.out_bf3 = 1;
            P.linear(ln3, q + "pwconv1", hid3, 4 * d, Bhw, 4 * d, 3 * d, eg);
            a3r_epilogue es = P.epi(A3R_EPI_RESID, nullptr, X);                                              // input + gamma * pwconv2(...) (gamma folded)
            es.aux_bf3 = S3;
            P.linear(hid3, q + "pwconv2", S, 3 * d, Bhw, 3 * d, 4 * d, es);
            P.linear(S3, q + "final", net, d, Bhw, d, 3 * d, P.epi(A3R_EPI_NONE, nullptr));
        }
        heads();
        P.launch([&](hipStream_t st) { hipLaunchKernelGGL(flow_add_kernel, dim3(grid1d(Bhw * 2)), dim3(256), 0, st, flow8, fu, 6, Bhw); });
        if (taps && it < 4) {
            tap(taps->net[it], net, (size_t)Bhw * d);
            tap(taps->flow8[it], flow8, (size_t)Bhw * 2);
        }
    }
    // flow_up = upsample_data(flow_8x, info_8x, weight_update)[0] (raft.py:183-199, 236)
    P.launch([&](hipStream_t st) { hipLaunchKernelGGL(convex_upsample_kernel, dim3(grid1d(Bhw * 64)), dim3(256), 0, st, flow8, wgt, flow_out, B, h, w); });
    if (peak) *peak = ar.peak;
    return P.rc;
}
}  // namespace

static int raft_check(a3r_raft_t m, int B, int H, int W, int iters, const char* who) {
    A3R_CHECK_ARG(m, "%s: null handle", who);
    A3R_CHECK_ARG(B > 0 && iters >= 0, "%s: batch must be positive, iters non-negative", who);
    // InputPadder (utils.py:11-28) pads to multiples of 8; the pyramid halves the 1/8 map corr_levels times (corr.py:22)
    A3R_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "%s: image size %dx%d must be a multiple of 8 (pad first, as InputPadder does)", who, H, W);
    A3R_CHECK_ARG(((H / 8) >> (m->cfg.corr_levels - 1)) >= 2 && ((W / 8) >> (m->cfg.corr_levels - 1)) >= 2,
                  "%s: image %dx%d too small for a %d-level correlation pyramid", who, H, W, m->cfg.corr_levels);
    return A3R_OK;
}

extern "C" size_t a3r_raft_workspace_bytes(a3r_raft_t m, int B, int H, int W) {
    if (!m || B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return 0;
    size_t peak = 0;
    raft_plan(m, true, nullptr, nullptr, B, H, W, 1, nullptr, nullptr, 0, nullptr, &peak, nullptr);
    return peak;
}

extern "C" int a3r_raft_set_arith(a3r_raft_t m, int fh2) {
    A3R_CHECK_ARG(m, "a3r_raft_set_arith: null handle");
    const int prev = m->use_fh2 ? 1 : 0;
    m->use_fh2 = fh2 != 0;
    return prev;
}

extern "C" int a3r_raft_range(a3r_raft_t m, float* max_abs_host, void* stream) {
    A3R_CHECK_ARG(m && max_abs_host, "a3r_raft_range: null argument");
    if (!m->finalized || !m->stat) { set_error("a3r_raft_range: a3r_raft_finalize has not been called"); return A3R_ESTATE; }
    A3R_HIP(hipMemcpyAsync(max_abs_host, m->stat, 4, hipMemcpyDeviceToHost, as_stream(stream)));
    A3R_HIP(hipStreamSynchronize(as_stream(stream)));
    return A3R_OK;
}

extern "C" int a3r_raft_encode(a3r_raft_t m, const float* image, int B, int H, int W, float* fmap, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (int rc = raft_check(m, B, H, W, 0, "a3r_raft_encode")) return rc;
    if (!m->finalized) { set_error("a3r_raft_encode: a3r_raft_finalize has not been called"); return A3R_ESTATE; }
    A3R_CHECK_ARG(image && fmap && workspace, "a3r_raft_encode: null pointer");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "a3r_raft_encode: workspace must be 256-byte aligned");
    size_t need_bytes = 0;
    raft_plan(m, true, nullptr, nullptr, B, H, W, 0, nullptr, nullptr, 0, nullptr, &need_bytes, nullptr, 1);
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_raft_encode: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return raft_plan(m, false, image, nullptr, B, H, W, 0, nullptr, workspace, workspace_bytes, stream, nullptr, nullptr, 1, nullptr, nullptr, fmap);
}

extern "C" int a3r_raft_forward_features(a3r_raft_t m, const float* image1, const float* image2, const float* fmap1, const float* fmap2, int B,
                                         int H, int W, int iters, float* flow, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = raft_check(m, B, H, W, iters, "a3r_raft_forward_features")) return rc;
    if (!m->finalized) { set_error("a3r_raft_forward_features: a3r_raft_finalize has not been called"); return A3R_ESTATE; }
    A3R_CHECK_ARG(image1 && image2 && fmap1 && fmap2 && flow && workspace, "a3r_raft_forward_features: null pointer");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "a3r_raft_forward_features: workspace must be 256-byte aligned");
    const size_t need_bytes = a3r_raft_workspace_bytes(m, B, H, W);
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_raft_forward_features: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return raft_plan(m, false, image1, image2, B, H, W, iters, flow, workspace, workspace_bytes, stream, nullptr, nullptr, 0, fmap1, fmap2);
}

extern "C" int a3r_raft_forward(a3r_raft_t m, const float* image1, const float* image2, int B, int H, int W, int iters, float* flow,
                                void* workspace, size_t workspace_bytes, const a3r_raft_taps* taps, void* stream) {
    if (int rc = raft_check(m, B, H, W, iters, "a3r_raft_forward")) return rc;
    if (!m->finalized) { set_error("a3r_raft_forward: a3r_raft_finalize has not been called"); return A3R_ESTATE; }
    A3R_CHECK_ARG(image1 && image2 && flow && workspace, "a3r_raft_forward: null pointer");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "a3r_raft_forward: workspace must be 256-byte aligned");
    const size_t need_bytes = a3r_raft_workspace_bytes(m, B, H, W);
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_raft_forward: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return raft_plan(m, false, image1, image2, B, H, W, iters, flow, workspace, workspace_bytes, stream, nullptr, taps);
}
