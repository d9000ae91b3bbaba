/*
 * a3r.h -- C ABI of liba3r.so, the MI355X (gfx950) pair-forward + global-alignment engine.
 *
 * The reference (OliEfr/Align3R) has no plugin/operator registry: its only native boundary is the
 * pybind function curope.rope_2d (croco/models/curope/curope.cpp:49-69); everything else on the hot
 * path is reached through Python.  This header therefore declares
 *   (1) a3r_rope2d            -- same semantics as curope.rope_2d,
 *   (2) the operators a reference nn.Module on the path maps to (Linear, LayerNorm, Attention,
 *       Conv2d/ConvTranspose2d/Interpolate of the DPT head, postprocess), each citing the reference
 *       lines it replaces, so that the Python mirror modules stay thin,
 *   (3) a3r_model_*           -- the whole AsymmetricCroCo3DStereo.forward (dust3r/model.py:241-257),
 *   (4) a3r_align_*           -- the PointCloudOptimizer inner loop (dust3r/cloud_opt/optimizer.py:223-241
 *                                + base_opt.py:424-464: loss, gradients, Adam).
 *
 * Conventions
 *   - plain C types only; every pointer is a CALLER-OWNED DEVICE buffer (e.g. torch tensor data_ptr())
 *     unless the name ends in _host; the library never allocates or frees device memory;
 *   - float = IEEE fp32; activations are channels-last: tokens [B, N, C], maps [B, H, W, C];
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously, nothing synchronises;
 *   - return value 0 = ok; otherwise a negative A3R_E* code and a3r_last_error() holds a message
 *     (the Python wrapper raises RuntimeError, mirroring TORCH_CHECK in curope.cpp:54-59);
 *   - handles are not thread-safe; use one handle per host thread / GPU.
 */
#ifndef A3R_H
#define A3R_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3R_OK 0
#define A3R_EINVAL (-1)   /* bad argument / shape */
#define A3R_EHIP (-2)     /* a HIP runtime call failed */
#define A3R_ESTATE (-3)   /* handle not in the required state (e.g. weights missing) */

const char* a3r_last_error(void);
int a3r_version(void);
/* number of HIP devices visible; does not initialise a context */
int a3r_device_count(void);

/* Per-kernel timing with HIP events recorded on the launch stream around every kernel launch of the
 * library (used by bench.py for the roofline numbers).  a3r_prof_enable(1) clears and starts recording,
 * a3r_prof_enable(0) stops.  a3r_prof_get synchronises the recorded events and returns, for kernel class
 * `kernel` (0 <= kernel < a3r_prof_kernel_count()), the number of launches, their summed duration and
 * the summed algorithmic work (FLOP for the MFMA kernels, bytes for the HBM-bound ones). */
int a3r_prof_enable(int on);
int a3r_prof_kernel_count(void);
int a3r_prof_get(int kernel, const char** name, long* launches, double* total_ms, double* total_work);
/* Summed ALGORITHMIC bytes of the MFMA-bound kernel classes (every operand read once, every result written once; 0 for classes
 * whose `work` already is a byte count): lets bench.py print algorithmic bytes per launch next to the PMC traffic. */
int a3r_prof_get_bytes(int kernel, double* total_bytes);

/* ------------------------------------------------------------------------------------------------
 * (1) 2-D rotary embedding, in place.  Replaces curope.rope_2d(tokens, positions, base, fwd)
 *     croco/models/curope/curope.cpp:49-69 / kernels.cu:17-108.
 *     tokens [B, N, H, D] fp32 (D % 4 == 0), positions [B, N, 2] int64 (y, x);
 *     first D/2 dims rotate with y, last D/2 with x, pairs (d, d + D/4), angle = fwd*pos*base^(-d/(D/4)).
 */
int a3r_rope2d(float* tokens, const int64_t* positions, int B, int N, int H, int D, float base, float fwd,
               void* stream);

/* ------------------------------------------------------------------------------------------------
 * (2) operators
 */

/* nn.LayerNorm over the last dim, eps as given (croco.py:34 uses 1e-6). x,y [M, D]; y may alias x. */
int a3r_layernorm(const float* x, const float* w, const float* b, float* y, int M, int D, float eps, void* stream);

/* Epilogues of a3r_linear / a3r_conv3x3 */
enum {
    A3R_EPI_NONE = 0,     /* y = acc (+bias) */
    A3R_EPI_GELU = 1,     /* y = gelu_erf(acc + bias)                      Mlp.fc1+act  blocks.py:73-75 */
    A3R_EPI_RESID = 2,    /* y = resid + acc + bias  (resid may alias y)  residual adds blocks.py:128-129 */
    A3R_EPI_RELU = 3,     /* y = relu(acc + bias) */
    A3R_EPI_ROPE = 4,     /* y = rope2d(acc + bias) on the first rope_cols columns (heads of 64), rest plain:
                             fuses RoPE2D (pos_embed.py:141-157) into the q/k projections blocks.py:96-103,155-162 */
    A3R_EPI_RESID2 = 5,   /* y = resid + resid2 + acc + bias              RCU skip + fusion add dpt_block.py:142,198 */
    A3R_EPI_PIXSHUF = 6,  /* ConvTranspose2d(kernel=stride=s) scatter: row = input pixel, col = (dy,dx,co) */
    A3R_EPI_HEAD = 7      /* fh2 kernels, N == 128 only: the tail of the DPT head fused into its last 3x3 conv -- t = relu(acc + bias)
                             [M, 128] never reaches memory; f = t @ head_w^T + head_b (Conv2d 1x1 128 -> 4, dpt_block.py:328) and the
                             postprocess (heads/postprocess.py:10-58: pts3d = xyz / max(|xyz|, 1e-8) * expm1(|xyz|), conf = 1 + exp(f3))
                             are done in the epilogue: y = pts3d [M, 3], head_conf = conf [M] */
};

typedef struct {
    int epi;              /* A3R_EPI_* */
    const float* bias;    /* [N] or NULL */
    const float* resid;   /* [M, ldc] for RESID/RESID2 */
    const float* resid2;  /* [M, ldc] for RESID2 */
    int relu_a;           /* a3r_conv3x3 only: relu on the input while staging it (RCU pre-activation dpt_block.py:131,136) */
    /* ROPE */
    int rope_cols;        /* leading output columns to rotate (multiple of 64) */
    int tokens_per_image; /* N tokens; row r has token index r % N */
    int grid_w;           /* tokens per image row: pos = (tok / grid_w, tok % grid_w)  (PositionGetter blocks.py:195-207) */
    const float* rope_cos;/* [max_pos, 16] cos table (a3r_rope_table) */
    const float* rope_sin;
    /* PIXSHUF */
    int ps_s, ps_h, ps_w, ps_cout;  /* stride s, input map h x w, output channels */
    /* a3r_linear_bf3 only: write y in bf3 form ([M][N/8][3][8] bf16, N % 8 == 0, ldc = N) instead of fp32, for outputs that
     * only feed the next bf3 kernel (Mlp: fc1 + GELU -> fc2, blocks.py:73-77; qkv + RoPE -> attention).  NONE / GELU / RELU / ROPE. */
    int out_bf3;
    /* bf3 kernels only: ALSO write the result (after bias / activation / residuals; through a ReLU if aux_relu) in bf3 form
     * [M][N/8][3][8] to aux_bf3 -- the pre-activated input of the next 3x3 conv of a ResidualConvUnit, whose fp32 value is
     * still needed for the skip connection (dpt_block.py:131-141). */
    void* aux_bf3;
    int aux_relu;
    /* a3r_linear_bf3 only -- operand layouts rather than epilogue options, kept here so the entry points stay as they are:
     * x_pair: x3 is stored in the ROW-PAIR form (the layout a3r_split_bf3_w documents; produced by a3r_split_bf3_w,
     * a3r_layernorm_bf3(pair = 1), a3r_attention_bf3(out_pair = 1) or an out_bf3 + out_pair epilogue).  out_pair: with out_bf3, write
     * y in that form (N % 32 == 0) -- for outputs only ever read as the x3 of another a3r_linear_bf3 (fc1 + GELU -> fc2).
     * A matrix with an odd number of rows occupies (rows + 1) * K * 6 bytes in this form (a3r_bf3_w_bytes). */
    int x_pair;
    int out_pair;
    /* a3r_linear_fh2 only: write y in fh2 form ([M][N/8][2][8] fp16, N % 32 == 0, ldc = N) instead of fp32, for outputs that only
     * feed the next a3r_linear_fh2 (Mlp: fc1 + GELU -> fc2, blocks.py:73-77).  NONE / GELU / RELU. */
    int out_fh2;
    /* fh2 kernels only: ALSO write the result (after bias / activation / residuals; through a ReLU if aux_relu) in fh2 form
     * [M][N/8][2][8] to aux_fh2 -- as aux_bf3, for the DPT convolutions on the fh2 kernel (y, resid, resid2 16-byte aligned). */
    void* aux_fh2;
    /* fh2 kernels only -- range control of the fh2 operands (see a3r_model_range_check).  All three may stay zero.
     * x_scale: the power of two the x2 operand was stored with (0 = 1); out_scale: the power of two to store the out_fh2 / aux_fh2
     * output with (0 = 1: the planes hold out_scale * value); out_absmax: device word that receives max |out_scale * value| over
     * that output by an atomic max on the bit pattern (NULL: no statistics; the caller zeroes it beforehand).  The grouped entry
     * point takes them per group (a3r_group_ptrs_fh2) and ignores these. */
    float x_scale, out_scale;
    unsigned* out_absmax;
    /* A3R_EPI_HEAD */
    const float* head_w;  /* [4, 128] */
    const float* head_b;  /* [4] */
    float* head_conf;     /* [M] */
    /* RESID / RESID2 only (fp32 y, LDS-staged or direct epilogue of the 16x16 kernels): y = relu(resid (+ resid2) + acc + bias) -- the
     * tail of a ResNet BasicBlock, `return self.relu(x + y)` (third_party/RAFT/core/layer.py:141); an aux_bf3 / aux_fh2 twin then holds the
     * same relu-ed value */
    int relu_out;
    /* RESID / RESID2 only: the branch is clamped BEFORE the addition, y = resid + relu(acc + bias) (then relu_out) -- BasicBlock's
     * `y = relu(bn2(conv2(y))); return relu(x + y)` (layer.py:135-141) */
    int relu_acc;
} a3r_epilogue;

/* nn.Linear: y[M, N] = x[M, K] @ w[N, K]^T (+ epilogue).  lda/ldc = row strides in floats
 * (K % 32 == 0, lda % 4 == 0).  fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 products. */
int a3r_linear(const float* x, int lda, const float* w, float* y, int ldc, int M, int N, int K,
               const a3r_epilogue* epi, void* stream);

/* Up to 4 same-shape nn.Linear problems in ONE launch (e.g. the two decoders' projections, model.py:218-220):
 * per-problem operands, shared M/N/K, leading dimensions and epilogue kind (epi->bias/resid/resid2 are ignored,
 * the per-group pointers are used instead). */
typedef struct {
    const float* x;
    const float* w;
    float* y;
    const float* bias;
    const float* resid;
    const float* resid2;
} a3r_group_ptrs;
int a3r_linear_grouped(const a3r_group_ptrs* groups, int n_groups, int lda, int ldc, int M, int N, int K,
                       const a3r_epilogue* epi, void* stream);

/* ---- the same nn.Linear on the bf16 matrix cores, fp32-accurate ("bf3" operands)
 * An fp32 matrix X [R, K] (K % 8 == 0) in bf3 form is three bf16 planes with X = X0 + X1 + X2 exactly, laid out
 * [R][K/8][3][8] bf16 (6 K bytes per row, a3r_bf3_bytes).  a3r_linear_bf3 evaluates the six bf16 x bf16 plane
 * products whose magnitude reaches fp32 precision (each exact in the fp32 accumulator; the dropped terms are
 * <= 2^-23 |x w| per product), so y matches a3r_linear to fp32 rounding while running on the 16x faster bf16 MFMA
 * pipes.  Replaces the same call sites as a3r_linear (blocks.py:58-169); epilogues as above. K % 32 == 0. */
size_t a3r_bf3_bytes(long rows, int K);
/* Process-wide arithmetic mode of the bf3 kernels (a3r_linear_bf3 / a3r_conv3x3_bf3 / a3r_attention_bf3): the plane products
 * evaluated per multiply.  6 (default): all products down to 2^-16, fp32-accurate.  3: a0 b0 + a0 b1 + a1 b0, operands
 * effectively 16 bits (error ~2^-17 per product).  1: a0 b0 only = plain bf16 operands, fp32 accumulation -- BASELINE config 5's
 * "bf16-MFMA mode", a reduced-precision mode that is never the default.  Returns the previous value; other values are ignored. */
int a3r_bf3_set_products(int products);
/* Process-wide arithmetic mode of the fh2 kernels (a3r_linear_fh2 / a3r_conv3x3_fh2 / a3r_attention_fh2): matrix passes per product.
 * 3 (default): h0 g0 + h0 g1 + h1 g0, fp32-grade.  1: h0 g0 only = plain fp16 operands (11 significant bits, kept inside fp16's range by
 * the same per-site scales), fp32 accumulation -- the 16-bit operand mode (A3R_GEMM=f16; BASELINE config 5's reduced-precision role with
 * three more mantissa bits than bf16), never the default.  Returns the previous value; other values are ignored. */
int a3r_fh2_set_passes(int passes);
/* fp32 x [M, ldx] (first K columns) -> bf3 y [M][K/8][3][8] */
int a3r_split_bf3(const float* x, int ldx, void* y, long M, int K, void* stream);
/* WEIGHT operands (w3 of a3r_linear_bf3, wp3 of a3r_conv3x3_bf3) use the row-pair form of the layout,
 * [ceil(N/2)][K/32][2 rows][4][3][8] bf16 (K % 32 == 0): rows 2j, 2j+1 interleaved per 32-deep k block, so the 32 k of a row
 * pair that one GEMM stage fetches are 384 contiguous bytes = three whole cache lines (the plain form drags 2 lines per row for
 * 1.5 lines of payload).  byte offset of (n, k) = (n/2) 12K + (k/32) 384 + (n%2) 192 + ((k/8)%4) 48 + plane 16 + (k%8) 2. */
size_t a3r_bf3_w_bytes(long rows, int K);
int a3r_split_bf3_w(const float* w, int ldw, void* y, long N, int K, void* stream);
/* nn.LayerNorm (as a3r_layernorm) writing its output directly in bf3 form (D % 8 == 0): the producer of every
 * transformer GEMM input (blocks.py:127-130,186-190), fused so the fp32 normalised rows never reach HBM.
 * pair != 0: y3 in the row-pair form (D % 32 == 0), for rows that only feed a3r_linear_bf3 (x_pair). */
int a3r_layernorm_bf3(const float* x, const float* w, const float* b, void* y3, int M, int D, float eps, int pair, void* stream);
/* x3: bf3 [M, K] (a3r_split_bf3 / a3r_layernorm_bf3 / a bf3 epilogue output); w3: [N, K] in the weight layout (a3r_split_bf3_w) */
int a3r_linear_bf3(const void* x3, const void* w3, float* y, int ldc, int M, int N, int K, const a3r_epilogue* epi,
                   void* stream);
typedef struct {
    const void* x3;       /* bf3 [M, K] */
    const void* w3;       /* bf3 [N, K], row-pair weight layout (a3r_split_bf3_w) */
    float* y;
    const float* bias;
    const float* resid;
    const float* resid2;
} a3r_group_ptrs_bf3;
int a3r_linear_bf3_grouped(const a3r_group_ptrs_bf3* groups, int n_groups, int ldc, int M, int N, int K,
                           const a3r_epilogue* epi, void* stream);

/* ---- the same nn.Linear on the fp16 matrix cores, fp32-GEMM-accurate ("fh2" operands)
 * An fp32 matrix X [R, K] (K % 32 == 0) in fh2 form is two fp16 planes, s X ~ X0 + X1 (X0 = rn_f16(s X), X1 = rn_f16(s X - X0)), laid
 * out [R][K/8][2][8] fp16 (4 K bytes per row, a3r_fh2_bytes), with s a power of two: 1 for activations, a3r_fh2_weight_scale(max|w|)
 * for weights.  a3r_linear_fh2 evaluates x0 w0 + x0 w1 + x1 w0 -- three fp16 MFMA passes, every product exact in the fp32
 * accumulator -- so the operands carry 22 of fp32's 24 significant bits; against float64 the result is as accurate as a3r_linear's
 * (the fp32 accumulation error dominates: tests/test_gpu_fh2.py) at half the matrix passes of a3r_linear_bf3.
 * Same call sites as a3r_linear (blocks.py:58-169); epilogues as above (no PIXSHUF), out_bf3 for the projections that feed the
 * attention kernel, out_fh2 (NONE / GELU / RELU / ROPE) for fc1 + GELU -> fc2 and for the q / k / v of a3r_attention_fh2. */
size_t a3r_fh2_bytes(long rows, int K);
/* fp32 x [M, ldx] (first K columns) * scale -> fh2 y (K % 8 == 0); absmax (device, may be NULL): atomic max of |scale * x| */
int a3r_split_fh2(const float* x, int ldx, void* y, long M, int K, float scale, unsigned* absmax, void* stream);
/* max |x| over n floats -> *out_dev (device float; the call zeroes it first) and the weight scale derived from it (host helper) */
int a3r_absmax(const float* x, long n, float* out_dev, void* stream);
float a3r_fh2_weight_scale(float absmax);
/* nn.LayerNorm (as a3r_layernorm) writing scale * y in fh2 form (D % 32 == 0; scale a power of two, absmax as a3r_split_fh2) */
int a3r_layernorm_fh2(const float* x, const float* w, const float* b, void* y2, int M, int D, float eps, float scale,
                      unsigned* absmax, void* stream);
typedef struct {
    const void* x2;       /* fh2 [M, K], stored with x_scale */
    const void* w2;       /* fh2 [N, K], stored with w_scale */
    float* y;
    const float* bias;
    const float* resid;
    const float* resid2;
    float w_scale;
    float x_scale;        /* 0 = 1 */
    float out_scale;      /* out_fh2 / aux_fh2 output: 0 = 1 */
    unsigned* out_absmax; /* or NULL */
} a3r_group_ptrs_fh2;
int a3r_linear_fh2_grouped(const a3r_group_ptrs_fh2* groups, int n_groups, int ldc, int M, int N, int K, const a3r_epilogue* epi,
                           void* stream);
int a3r_linear_fh2(const void* x2, const void* w2, float w_scale, float* y, int ldc, int M, int N, int K, const a3r_epilogue* epi,
                   void* stream);

/* nn.Conv2d(k=3, padding=1, stride in {1,2}) as an implicit GEMM on the bf3 kernel: x3 = bf3 form of the channels-last map
 * [B, H, W, Cin] (i.e. of the [B H W, Cin] matrix), wp3 = bf3 form of the packed weights [Cout, 9 Cin] (a3r_pack_conv3x3
 * then a3r_split_bf3_w); y [B, Ho, Wo, Cout] fp32 (or bf3 with out_bf3).  Same call sites as a3r_conv3x3 (dpt_block.py). */
int a3r_conv3x3_bf3(const void* x3, const void* wp3, float* y, int B, int H, int W, int Cin, int Cout, int stride,
                    const a3r_epilogue* epi, void* stream);

/* The same on the fh2 kernel: x2 = fh2 form of the channels-last map (scale 1), wp2 = fh2 form of the packed weights [Cout, 9 Cin]
 * (a3r_pack_conv3x3 then a3r_split_fh2 with the power-of-two w_scale of a3r_fh2_weight_scale); Cin % 32 == 0.  y fp32
 * (optionally also aux_fh2) or fh2 (out_fh2, Cout % 32 == 0).  Same call sites as a3r_conv3x3 (dpt_block.py:120-142, 186-218, 323-330). */
int a3r_conv3x3_fh2(const void* x2, const void* wp2, float w_scale, float* y, int B, int H, int W, int Cin, int Cout, int stride,
                    const a3r_epilogue* epi, void* stream);

/* nn.Conv2d(k=3, padding=1, stride in {1,2}) on channels-last x [B, H, W, Cin] with PACKED weights
 * wp [Cout, 3, 3, Cin] (a3r_pack_conv3x3 from the checkpoint layout [Cout, Cin, 3, 3]); Cin % 32 == 0.
 * y [B, Ho, Wo, Cout].  Implicit GEMM on the same MFMA core.  (dpt_block.py:33-68,93-111,323-329,402-405) */
int a3r_conv3x3(const float* x, const float* wp, float* y, int B, int H, int W, int Cin, int Cout, int stride,
                const a3r_epilogue* epi, void* stream);
int a3r_pack_conv3x3(const float* w, float* wp, int Cout, int Cin, void* stream);
/* ConvTranspose2d weight [Cin, Cout, s, s] -> [(dy*s+dx)*Cout + co, Cin] for a3r_linear + A3R_EPI_PIXSHUF */
int a3r_pack_convT(const float* w, float* wp, int Cin, int Cout, int s, void* stream);

/* softmax(q k^T / sqrt(64)) v per head, head_dim 64 (Attention/CrossAttention blocks.py:105-109,164-168).
 * q [B, Nq, ldq], k/v [B, Nk, ldk/ldv] with head h at column h*64; o [B, Nq, ldo].  q,k already rotated. */
int a3r_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                  int B, int H, int Nq, int Nk, void* stream);
/* The same attention on the bf16 matrix cores with fp32 accuracy: q3 / k3 / v3 / o3 are bf3 matrices (pointers to the first
 * column's 48-byte group, leading dimensions in fp32 columns, multiples of 8); six exact bf16 MFMA passes per product,
 * fp32 softmax, P split exactly into three planes in registers.  Consumes the RoPE + out_bf3 output of a3r_linear_bf3 and
 * produces the bf3 input of the output projection (out_pair != 0: in the row-pair form, ldo % 32 == 0, o3 = the matrix origin;
 * q3 / k3 / v3 are always plain rows). */
int a3r_attention_bf3(const void* q3, int ldq, const void* k3, int ldk, const void* v3, int ldv, void* o3, int ldo,
                      int B, int H, int Nq, int Nk, int out_pair, void* stream);
/* the same, writing o in fh2 form (scale 1, plain rows, ldo % 8 == 0): the input of an a3r_linear_fh2 output projection */
int a3r_attention_bf3_fh2out(const void* q3, int ldq, const void* k3, int ldk, const void* v3, int ldv, void* o2, int ldo,
                             int B, int H, int Nq, int Nk, void* stream);

/* The same attention on the fp16 matrix cores from fh2 operands (q2 / k2 / v2 / o2: fh2 matrices, pointers to the first column's
 * 32-byte group, leading dimensions in fp32 columns, multiples of 8): three exact fp16 MFMA passes per product, fp32 softmax,
 * P split into two fp16 planes of 1024 p in registers.  Consumes the RoPE + out_fh2 output of a3r_linear_fh2 and produces the fh2
 * input of the output projection.  range (or NULL = all ones, no statistics): the powers of two q2 / k2 / v2 were stored with (divided
 * out exactly: q_scale k_scale in the exp2 argument, v_scale with the softmax normaliser), the one to store o2 with, and the
 * device word receiving max |out_scale * o| (see a3r_epilogue.out_absmax); zeros mean 1 / none. */
typedef struct {
    float q_scale, k_scale, v_scale, out_scale;
    unsigned* out_absmax;
} a3r_fh2_attn_range;
int a3r_attention_fh2(const void* q2, int ldq, const void* k2, int ldk, const void* v2, int ldv, void* o2, int ldo,
                      int B, int H, int Nq, int Nk, const a3r_fh2_attn_range* range, void* stream);
/* Kernel form behind a3r_attention_fh2: 2 (default: K and V tiles by LDS-DMA, transposed LDS reads) or 1 (round 2: V staged through
 * registers); the results are bitwise equal (tests/test_gpu_fh2.py).  Returns the previous form, or a negative error code.
 * Process-wide; environment A3R_ATTN=v1 selects form 1 at start-up.  A development switch, not part of the reference's interface. */
int a3r_attention_fh2_set_form(int form);

/* cos/sin tables [max_pos, 16] for head_dim 64 computed like RoPE2D.get_cos_sin (pos_embed.py:118-128);
 * HOST buffers. */
int a3r_rope_table_host(float* cos_host, float* sin_host, int max_pos, float base);

/* PatchEmbedDust3R conv k=s=16 as im2col: cols[B*N, C*256] with k = c*256 + py*16 + px
 * (patch_embed.py:19-29).  img element (b,c,y,x) at b*sb + c*sc + y*sy + x*sx (floats). */
int a3r_patchify(const float* img, float* cols, int B, int C, int H, int W, long sb, long sc, long sy, long sx,
                 void* stream);

/* F.interpolate(scale_factor=2, bilinear, align_corners=True) on [B, H, W, C], writing only the
 * top-left Hc x Wc window of the 2H x 2W result (crop of dpt_head.py:57 folded in). C % 4 == 0. */
/* Raw weighted second moments of B point-set pairs for the similarity registrations of the aligner's initialisation
 * (cloud_opt/init_im_poses.py:415-418, roma.rigid_points_registration): problem b uses x + x_off[b], y + y_off[b] ([P,3] each) and
 * w + w_off[b] ([P]) (element offsets, device int64).  partial [B][a3r_umeyama_chunks(P)][17] doubles:
 * sum w | sum w x (3) | sum w y (3) | sum w |x|^2 | sum w y_r x_c (9, r major); the caller adds the chunks. */
int a3r_umeyama_chunks(int P);
int a3r_umeyama_moments(const float* x, const float* y, const float* w, const long* x_off, const long* y_off, const long* w_off,
                        int B, int P, double* partial, void* stream);

/* Closed-form weighted similarity registration of B problems from the raw moments of a3r_umeyama_moments (partial [B][nch][17]):
 * out [B][13] = (s, R row-major [9], T [3]) minimising sum w |s R x + T - y|^2 -- 3x3 SVD by Jacobi on the device, one thread per
 * problem (replaces roma.rigid_points_registration, init_im_poses.py:415-418). */
int a3r_umeyama_solve(const double* partial, int nch, int B, float* out, void* stream);
/* Batched camera pose from a world-space point map with known intrinsics (stands in for fast_pnp, init_im_poses.py:442-482; parity
 * unpinned: cv2.solvePnPRansac is stochastic and absent).  desc: B device records of a3r_pnp_desc_bytes() bytes
 * {const float* pts [H,W,3]; const uint8* mask [H,W]; int H, W, step, n; float focal, ppx, ppy, pad} -- the problem uses the n =
 * ceil(H W / step) pixels p * step that the mask keeps.  Closed-form start + `iterations` robust Gauss-Newton steps on the reprojection error
 * (Cauchy weights annealed to 5 px), float64, no host synchronisation.  c2w [B][16] camera-to-world;
 * info [B][4] = (valid, inliers < 5 px in front of the camera, sum of min(e^2, 25 px^2) over the points in front, focal).  work: a3r_pnp_work_bytes(B, n_max). */
size_t a3r_pnp_desc_bytes(void);
int a3r_pnp_chunks(int n_max);
size_t a3r_pnp_work_bytes(int B, int n_max);
int a3r_pnp_solve(const void* desc, int B, int n_max, int iterations, void* work, float* c2w, float* info, void* stream);

/* ---- per-pixel passes of the aligner's construction and of the MST initialisation (csrc/init_maps.hip), so that neither moves
 * the confidences through the host nor depends on which torch kernels happen to be loaded already.
 * a3r_conf_prepare: w_* [E, P] = trf(conf_* [E, P]) with mode 0 log | 1 sqrt | 2 x - 1 | 3 identity (commons.py:42-55; w_i = w_j =
 *   NULL skips it) and edge_mean [2 E] (or NULL): mean(conf_i[e]) at 2 e, mean(conf_j[e]) at 2 e + 1 (commons.py:20-25), float64
 *   sums in a fixed order.  16-byte aligned maps, P % 4 == 0.
 * a3r_im_conf_max: out [N, P] = per-image confidence = max over the edges an image appears in, from 0 (base_opt.py:169-175);
 *   ei / ej: DEVICE int32 [E].
 * a3r_weiszfeld_focal: focal [B] of B point maps [B, H, W, 3], principal point at the centre, `iterations` re-weighting steps
 *   (post_process.py:36-60, 'weiszfeld', 10 iterations in the reference), clipped at 0.
 * a3r_sim3_apply: y [P, 3] = post * (k R x + T) with (s, R row-major, T) = sol[0..12] in DEVICE memory (a row of a3r_umeyama_solve's
 *   output), k = s if with_scale else 1: geotrf(sRT_to_4x4(...), pts) without a host round trip (init_im_poses.py:226-233).
 * a3r_depth_init: depth [N, P] = log of the camera-space depth of scale * pts [N, P, 3] under the world-to-camera matrices w2c
 *   [N, 3, 4] (DEVICE), with log(z <= 0 | nan) -> 0 and log(inf) -> FLT_MAX, i.e. _set_depthmap's .log().nan_to_num(neginf=0)
 *   (init_im_poses.py:116-126).
 * a3r_mask_gt: out [n] (uint8) = x > thr. */
int a3r_conf_prepare(const float* conf_i, const float* conf_j, int E, long P, int mode, float* w_i, float* w_j, float* edge_mean,
                     void* stream);
int a3r_im_conf_max(const float* conf_i, const float* conf_j, const int* ei, const int* ej, int E, int N, long P, float* out,
                    void* stream);
int a3r_weiszfeld_focal(const float* pts3d, int B, int H, int W, int iterations, float* focal, void* stream);
int a3r_sim3_apply(const float* x, const float* sol, int with_scale, float post, float* y, long P, void* stream);
int a3r_depth_init(const float* pts, const float* w2c, float scale, int N, long P, float* depth, void* stream);
int a3r_mask_gt(const float* x, float thr, unsigned char* out, long n, void* stream);

int a3r_upsample2x(const float* x, float* y, int B, int H, int W, int C, int Hc, int Wc, void* stream);
/* the same, written in bf3 form ([B Hc Wc][C/8][3][8] bf16, C % 8 == 0): input of the next conv on the bf3 kernel */
int a3r_upsample2x_bf3(const float* x, void* y3, int B, int H, int W, int C, int Hc, int Wc, void* stream);
/* the same, written in fh2 form ([B Hc Wc][C/8][2][8] fp16 planes of scale * y, C % 8 == 0): input of the next conv on the fh2
 * kernel; scale a power of two, absmax (device, may be NULL) receives max |scale * y| as in a3r_split_fh2 */
int a3r_upsample2x_fh2(const float* x, void* y2, int B, int H, int W, int C, int Hc, int Wc, float scale, unsigned* absmax,
                       void* stream);

/* last 1x1 conv (128 -> 4) + postprocess (dpt_block.py:329, postprocess.py:10-58):
 * x [P, C] -> pts3d [P, 3] = xyz/max(|xyz|,1e-8)*expm1(|xyz|), conf [P] = 1 + exp(c). */
int a3r_head_final(const float* x, const float* w, const float* b, float* pts3d, float* conf, long P, int C,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * (3) whole-model forward: AsymmetricCroCo3DStereo (dust3r/model.py:65-257)
 */
typedef struct {
    int enc_embed_dim, enc_depth, enc_num_heads;
    int dec_embed_dim, dec_depth, dec_num_heads;
    int mlp_ratio, patch_size;
    float rope_base;
    int feature_dim, last_dim;
    int layer_dims[4];
} a3r_model_config;

typedef struct a3r_model_s* a3r_model_t;

int a3r_model_create(const a3r_model_config* cfg, a3r_model_t* out);
int a3r_model_destroy(a3r_model_t m);
/* Register one checkpoint tensor by its state-dict name (reference layout, device pointer, fp32).
 * The pointer must stay valid for the lifetime of the handle. */
int a3r_model_set_weight(a3r_model_t m, const char* name, const float* ptr, int ndim, const int64_t* shape);
/* Bytes of device memory the caller must provide for repacked conv / fused-projection weights. */
size_t a3r_model_packed_bytes(a3r_model_t m);
/* Checks that every parameter is present, repacks into `packed` (caller-owned, >= packed_bytes). */
int a3r_model_finalize(a3r_model_t m, void* packed, size_t packed_bytes, void* stream);
/* Workspace bytes for a batch of B pairs of H x W images. */
size_t a3r_model_workspace_bytes(a3r_model_t m, int B, int H, int W);
/* One forward over B pairs.  img1/img2 [B,3,H,W]; pd1/pd2 = view['pred_depth'] [B,H,W,3];
 * outputs pts1 [B,H,W,3], conf1 [B,H,W], pts2 (= pts3d_in_other_view), conf2. */
int a3r_model_forward(a3r_model_t m, const float* img1, const float* img2, const float* pd1, const float* pd2,
                      int B, int H, int W, float* pts1, float* conf1, float* pts2, float* conf2,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Encoder feature caching (an algorithmic saving the reference does not have: it re-encodes each frame once per
 * pair, dust3r/inference.py:66 + model.py:165-174, although the encoder output depends on the frame only).
 * a3r_model_encode: B frames img [B,3,H,W] -> enc_norm'd tokens feat_out [B, (H/16)*(W/16), enc_embed_dim].
 * a3r_model_decode: the rest of forward() for B pairs from cached features feat1/feat2 [B, N, enc_embed_dim]
 * (workspace >= a3r_model_workspace_bytes).  encode + decode is bit-identical to a3r_model_forward. */
size_t a3r_model_encode_workspace_bytes(a3r_model_t m, int B, int H, int W);
int a3r_model_encode(a3r_model_t m, const float* img, int B, int H, int W, float* feat_out, void* workspace,
                     size_t workspace_bytes, void* stream);
int a3r_model_decode(a3r_model_t m, const float* feat1, const float* feat2, const float* pd1, const float* pd2, int B,
                     int H, int W, float* pts1, float* conf1, float* pts2, float* conf2, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Debug taps for the parity tests: intermediate tensors inside the workspace after a forward; returns device pointer + element
 * count.  name: "feat" (enc_norm output, model.py:163), "hook_a" / "hook_b" (the decoder levels the DPT head reads,
 * dpt_head.py:101), "dec_last" (dec_norm output, model.py:231-232); after a3r_model_set_tap_level(m, L > 0) also "level" (the two
 * decoders' outputs of block L incl. the zero-conv add, model.py:218-228) and "pc0" (patch_embed_point_cloud output,
 * model.py:244-248) -- two more [2 B N, dec_embed_dim] buffers in the workspace (a3r_model_workspace_bytes accounts for them). */
int a3r_model_set_tap_level(a3r_model_t m, int level);
int a3r_model_tap(a3r_model_t m, const char* name, const float** ptr, size_t* count);

/* Range control of the default (fh2) arithmetic.  The reference computes in fp32 (croco.py:13 even allows TF32) and has no
 * activation range limit; two fp16 planes do: a tensor is fp32-grade only while its largest |scale * x| lies in [2^-2, 2^15]
 * (csrc/fh2.h).  Every site of the launch plan that writes an fh2 tensor therefore carries its own power-of-two scale (initially 1)
 * and records max |scale * x| during a3r_model_forward / _encode / _decode.  a3r_model_range_check waits for `stream`, reads the
 * statistics of the LAST such call, and gives every site whose maximum left the band (or was not finite) a new scale that puts it at
 * [2^11, 2^12).  *n_adjusted = number of sites changed; *n_nonfinite = sites whose maximum was Inf / NaN (their own scale cannot be
 * derived until the upstream overflow is gone).  When either is non-zero the outputs of that call are NOT fp32-grade (an overflow
 * makes them Inf / NaN) and the call must be repeated -- the scales persist in the handle, so a checkpoint with large activations
 * pays this once.  With both zero the outputs are final.  Modes other than fh2 (A3R_GEMM=f32|bf3|...) have fp32 range: always 0.
 * a3r_model_reset_ranges puts every scale back to 1. */
int a3r_model_range_check(a3r_model_t m, void* stream, int* n_adjusted, int* n_nonfinite);
int a3r_model_reset_ranges(a3r_model_t m);
/* (diagnostic) max |scale * x| every site of the LAST forward / encode / decode call recorded, in plan order (waits for `stream`);
 * 0 = all-zero tensor or a site that did not run. */
int a3r_model_range_stats(a3r_model_t m, void* stream, float* stored_absmax, int capacity, int* n_sites);
/* (diagnostic) the current scales of plan `phase` (0 = forward, 1 = encode, 2 = decode), in plan order: *n_sites = their number,
 * the first min(capacity, *n_sites) are copied to scales (host). */
int a3r_model_range_scales(a3r_model_t m, int phase, float* scales, int capacity, int* n_sites);

/* ------------------------------------------------------------------------------------------------
 * (3b) RAFT2 ("SEA-RAFT") optical flow: the flow provider of cloud_opt_flow (dust3r/cloud_opt_flow/optimizer.py:118-154 ->
 *      third_party/raft.py:39-73 -> third_party/RAFT/core/raft.py:152-246), csrc/raft.hip.
 *      Weights: the reference's state_dict names with evaluation-mode BatchNorm folded into the convolution in front of it, the
 *      ConvNeXt layer scale `gamma` folded into pwconv2, the factor 0.25 of raft.py:216 folded into upsample_weight.2, and
 *      update_block.encoder.convc1's input channels zero-padded to a multiple of 32 (align3r_amd/raft_weights.py does all four).
 */
typedef struct {
    int initial_dim;       /* 64 */
    int block_dims[3];     /* 64, 128, 256 */
    int n_blocks[3];       /* 3, 4, 6 ('resnet34', extractor.py:286) */
    int dim;               /* 128: hidden = context = dim, feature maps 2 dim */
    int radius;            /* 4 */
    int corr_levels;       /* 4 (raft.py:159) */
    int num_blocks;        /* 2 ConvNeXt refinement blocks */
} a3r_raft_config;
typedef struct a3r_raft_s* a3r_raft_t;
/* optional copies of intermediates for the parity tests (device pointers or NULL), channels-last, h = H / 8, w = W / 8 */
typedef struct {
    float* cnet;           /* [B, h, w, 2 dim]  init_conv(cnet(cat(image1, image2)))           raft.py:207-209 */
    float* fmap;           /* [2 B, h, w, 2 dim] fnet(image1) then fnet(image2)                  raft.py:222-223 */
    float* corr_pyr[4];    /* [B h w, h_l, w_l]  the correlation pyramid                         corr.py:17-23 */
    float* flow_update0;   /* [B, h, w, 6]       flow_head(net) before the first iteration       raft.py:213 */
    float* weight0;        /* [B, h, w, 576]     .25 * upsample_weight(net), same point          raft.py:214 */
    float* lookup0;        /* [B, h, w, ceil32(levels (2r+1)^2)] the first correlation lookup    raft.py:228 */
    float* motion0;        /* [B, h, w, dim]     BasicMotionEncoder2's output, first iteration   update.py:108-117 */
    float* net[4];         /* [B, h, w, dim]     hidden state after iterations 0..3 */
    float* flow8[4];       /* [B, h, w, 2]       coarse flow after iterations 0..3 */
} a3r_raft_taps;
int a3r_raft_create(const a3r_raft_config* cfg, a3r_raft_t* out);
int a3r_raft_destroy(a3r_raft_t m);
int a3r_raft_set_weight(a3r_raft_t m, const char* name, const float* ptr, int ndim, const int64_t* shape);
size_t a3r_raft_packed_bytes(a3r_raft_t m);
int a3r_raft_finalize(a3r_raft_t m, void* packed, size_t packed_bytes, void* stream);
size_t a3r_raft_workspace_bytes(a3r_raft_t m, int B, int H, int W);
/* flow [B, 2, H, W] = RAFT2(image1, image2, iters, test_mode=True)[1]: image* [B, 3, H, W] with values in [0, 255] (the reference
 * feeds img * 255, optimizer.py:141-146); H, W multiples of 8 (the reference pads to that with InputPadder) and at least
 * 16 << (corr_levels - 1) (its pyramid needs it, corr.py:22). */
int a3r_raft_forward(a3r_raft_t m, const float* image1, const float* image2, int B, int H, int W, int iters, float* flow,
                     void* workspace, size_t workspace_bytes, const a3r_raft_taps* taps, void* stream);
/* Arithmetic of the flow network's convolutions / linear layers / correlation: 1 (default) = the two-plane fp16 form (fh2 kernels: three
 * fp16 MFMA passes, operands stored with scale 1), 0 = the three-plane bf16 form (six passes, fp32 range).  The fh2 form is fp32-grade
 * while every stored activation stays inside fp16's range; every fh2 producer of a call reports max |stored value| into one device
 * word that a3r_raft_range reads back (it waits for the stream): a caller that finds it at or above 2^15 (or NaN) repeats the call
 * after a3r_raft_set_arith(m, 0) -- align3r_amd.raft.RaftEngine does.  set_arith returns the previous setting. */
int a3r_raft_set_arith(a3r_raft_t m, int fh2);
int a3r_raft_range(a3r_raft_t m, float* max_abs_host, void* stream);
/* The feature network alone: fmap [B, H/8, W/8, 2 dim] = fnet(2 image / 255 - 1) for B frames (raft.py:222-223).  A frame's features do
 * not depend on the pair it appears in: the reference's cloud_opt_flow re-encodes every frame for every edge and direction
 * (optimizer.py:141-146); computing them once per frame and passing them to a3r_raft_forward_features gives the same flow with a
 * third fewer FLOPs.  workspace: a3r_raft_workspace_bytes(m, B, H, W) is sufficient. */
int a3r_raft_encode(a3r_raft_t m, const float* image, int B, int H, int W, float* fmap, void* workspace, size_t workspace_bytes, void* stream);
/* a3r_raft_forward with the two feature maps given (fmap1 / fmap2 [B, H/8, W/8, 2 dim] from a3r_raft_encode of image1 / image2). */
int a3r_raft_forward_features(a3r_raft_t m, const float* image1, const float* image2, const float* fmap1, const float* fmap2, int B, int H,
                              int W, int iters, float* flow, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (4) global alignment inner loop: PointCloudOptimizer (cloud_opt/optimizer.py, base_opt.py)
 */
typedef struct {
    int E, N, P;                 /* edges, images, max pixels per image */
    int use_mono;                /* depth = mono*exp(scalemap)+shift (optimizer.py:181-185) else exp(log-depth) */
    int norm_pw_scale;           /* base_opt.py:212-218 */
    int dist_l2;                 /* 0 = l1_dist, 1 = l2_dist (commons.py:102-107) */
    int train_poses, train_focals, train_pp;
    float base_scale, pw_break, focal_break;
    double total_area_i, total_area_j;     /* optimizer.py:70-71 */
    /* graph (HOST arrays, copied at create) */
    const int32_t* ei_host;      /* [E] */
    const int32_t* ej_host;      /* [E] */
    const int32_t* imw_host;     /* [N] image widths  */
    const int32_t* imarea_host;  /* [N] h*w; pixels >= imarea are padding (grid 0, weight 0) */
    /* stacked observations (device), optimizer.py:60-67 */
    const float* pred_i;         /* [E, P, 3] */
    const float* pred_j;         /* [E, P, 3] */
    const float* w_i;            /* [E, P] conf_trf(conf) */
    const float* w_j;            /* [E, P] */
    const float* mono;           /* [N, P] or NULL */
    const float* pp0;            /* [N, 2] = (w/2, h/2) */
    /* parameters (device, updated in place) */
    float* pw_poses;             /* [E, 8] quat xyzw, signed_log1p(T), log scale */
    float* pw_adaptors;          /* [E, 2] (frozen: allow_pw_adaptors=False) */
    float* depth;                /* [N, P] log-depth, or scalemap when use_mono */
    float* shifts;               /* [N] (use_mono) */
    float* im_poses;             /* [N, 7] */
    float* im_focals;            /* [N] focal_break*log(f) */
    float* im_pp;                /* [N, 2] */
    /* Adam state (device, zero-initialised by the caller): [m | v] for each parameter, same shapes */
    float* adam_pw_poses;        /* [2, E, 8] */
    float* adam_depth;           /* [2, N, P] */
    float* adam_small;           /* [2, N, 16]: per image (im_poses 7, im_focals 1, im_pp 2, shift 1, pad) */
    /* scratch + outputs (device) */
    void* workspace;             /* >= a3r_align_workspace_bytes(E, N, P) */
    size_t workspace_bytes;
    float* loss_history;         /* [loss_capacity]; entry t = loss of iteration t (before its update) */
    int loss_capacity;
    /* allow_pw_adaptors=True (base_opt.py:117-118,177-182): pw_adaptors [E,2] = (xy, z) log-scale adaptation of each pairwise
     * prediction become trainable: gradient + Adam like every other small parameter.  adam_pw_adaptors [2, E, 2] (zero-initialised)
     * is required when train_adaptors != 0 and ignored otherwise. */
    int train_adaptors;
    float* adam_pw_adaptors;
} a3r_align_desc;

typedef struct a3r_align_s* a3r_align_t;

size_t a3r_align_workspace_bytes(int E, int N, int P);
int a3r_align_create(const a3r_align_desc* desc, a3r_align_t* out, void* stream);
int a3r_align_destroy(a3r_align_t a);
/* One global_alignment_iter (base_opt.py:450-464): loss + gradients + Adam(betas .9,.9, eps 1e-8) with
 * learning rate lr.  Fully asynchronous; the loss lands in loss_history[step]. */
int a3r_align_step(a3r_align_t a, float lr, void* stream);
/* n iterations in one call: lrs_host[k] is the learning rate of iteration k (host array; the caller evaluates its schedule),
 * epochs first_epoch .. first_epoch + n - 1.  Same results as n a3r_align_step_epoch calls, without a host round trip each. */
int a3r_align_run(a3r_align_t a, const float* lrs_host, int n, int first_epoch, void* stream);
/* Loss only (net() without backward), written to *loss_dev (device float). */
int a3r_align_loss(a3r_align_t a, float* loss_dev, void* stream);
/* Gradients of the current state without updating (for the parity tests):
 * g_pw_poses [E,8], g_depth [N,P], g_small [N,16] (layout of adam_small), loss_dev [1]. */
int a3r_align_grad(a3r_align_t a, float* g_pw_poses, float* g_depth, float* g_small, float* loss_dev, void* stream);
/* the same with the gradient of pw_adaptors [E,2] as well (g_pw_adaptors may be NULL); epoch as a3r_align_grad_epoch */
int a3r_align_grad_full(a3r_align_t a, int epoch, float* g_pw_poses, float* g_pw_adaptors, float* g_depth, float* g_small,
                        float* loss_dev, void* stream);
/* cloud_opt_flow variant (dust3r/cloud_opt_flow/optimizer.py:36-116,500-572): shared focal, temporal smoothing of
 * consecutive image poses (relative_pose_loss), ego-flow smooth-L1 term against precomputed optical flow (the RAFT
 * fields and the dynamic masks are INPUTS).  Call once after a3r_align_create (not with use_mono).  With shared_focal
 * im_focals has one element and the Adam moments of the focal live in slot [0][7] of adam_small. */
typedef struct {
    int shared_focal;
    float temporal_smoothing_weight;   /* 0 disables */
    float translation_weight;
    float flow_loss_weight;            /* 0 disables the ego-flow term */
    float flow_loss_thre;              /* > 0: the term is dropped in an iteration whose flow loss exceeds it (optimizer.py:538) */
    float pxl_thre;                    /* per-element threshold of smooth_L1_loss_fn (optimizer.py:18-24) */
    int flow_start_iter;               /* first epoch e with e >= num_total_iter * flow_loss_start_epoch */
    int H, W;                          /* image shape, H*W == P */
    const float* flow_ij;              /* [E, 2, H*W] flow of image ei towards ej (x then y planes), device */
    const float* flow_ji;              /* [E, 2, H*W] */
    const uint8_t* dynamic_mask;       /* [N, H*W], 1 = dynamic pixel (excluded), device */
    void* workspace;                   /* >= a3r_align_flow_workspace_bytes(E, N, P), device */
    size_t workspace_bytes;
} a3r_align_flow_desc;
size_t a3r_align_flow_workspace_bytes(int E, int N, int P);
int a3r_align_set_flow(a3r_align_t a, const a3r_align_flow_desc* f, void* stream);
/* Depth prior of the flow variant: depth_regularize_weight * depth_regularization_si_weighted(depthmaps, init_depthmaps,
 * dynamic_masks) (dust3r/utils/goem_opt.py:15-36 as called at dust3r/cloud_opt_flow/optimizer.py:546-555): a scale-invariant
 * squared log-depth distance to the depth maps captured by _set_init_depthmap (optimizer.py:452-454), pixels of the dynamic
 * mask weighted 2, others 1, averaged over images.  init_log_depth: [N, P] log-depth parameters at capture time, device;
 * dynamic_mask: [N, P] (1 = dynamic) or NULL; both stay owned by the caller and must outlive the handle.  weight == 0 turns
 * the term off (other arguments ignored).  Loss, gradient export and step all include the term afterwards. */
size_t a3r_align_depth_prior_workspace_bytes(int N, int P);
int a3r_align_set_depth_prior(a3r_align_t a, float weight, const float* init_log_depth, const uint8_t* dynamic_mask,
                              void* workspace, size_t workspace_bytes, void* stream);
/* a3r_align_step with an explicit epoch (net(epoch=cur_iter), cloud_opt_flow/base_opt.py:571); a3r_align_step uses
 * the handle's own iteration count. */
int a3r_align_step_epoch(a3r_align_t a, float lr, int epoch, void* stream);
int a3r_align_grad_epoch(a3r_align_t a, int epoch, float* g_pw_poses, float* g_depth, float* g_small, float* loss_dev,
                         void* stream);
/* HOST out[5] = {c_ij, c_ji, last flow loss, dropped in the last evaluation, dropped ever (flow_loss_flag)}; synchronises. */
int a3r_align_flow_state(a3r_align_t a, float* state_host5);
int a3r_align_steps_done(a3r_align_t a);
/* Tell the handle that the caller rewrote parameter buffers (preset_pose, init, load_state_dict ...). */
int a3r_align_invalidate(a3r_align_t a);
/* per-edge [E,3,4] = [s*R*diag(a) | s*T] and per-image [N,3,4] = [R | t] (get_pw_poses/get_im_poses rows 0-2) */
int a3r_align_pose_matrices(a3r_align_t a, float* edge_M, float* img_R, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* A3R_H */
