// AsymmetricCroCo3DStereo.forward as a fixed launch plan over liba3r's kernels (host code only).
//
// Mirrors, step for step (all under /root/reference/):
//   forward / _encode_image_pairs / _encode_image   dust3r/model.py:241-257,151-174
//   _decoder (+ dec_blocks_pc / zero_convs branch)  dust3r/model.py:201-233
//   Block / DecoderBlock                             croco/models/blocks.py:114-191
//   DPTOutputAdapter_fix.forward                     dust3r/heads/dpt_head.py:34-66
//   postprocess                                      dust3r/heads/postprocess.py:10-58
// Data layout: tokens [rows, C] with rows = (side, batch, token): the first B*N rows belong to view 1,
// the next B*N rows to view 2 (this is the reference's torch.cat((img1, img2)) batch order), maps are
// channels-last.  All buffers live in the caller's workspace; nothing here allocates device memory.
#include "common.h"
#include <map>
#include <string>
#include <vector>
#include <new>
#include <cstdlib>
#include <cmath>
#include <cstring>

namespace a3r {

struct WRef { const float* p = nullptr; std::vector<int64_t> shape; };

struct Arena {
    char* base; size_t off, cap; bool dry; size_t peak; bool overflow = false;
    float* alloc(size_t nfloat) {
        size_t o = off;
        off = align_up(off + nfloat * 4, 256);
        if (off > peak) peak = off;
        if (!dry && off > cap) { overflow = true; return reinterpret_cast<float*>(base); }
        return dry ? nullptr : reinterpret_cast<float*>(base + o);
    }
};

struct BlockW {
    const float *n1w, *n1b, *qkvw, *qkvb, *projw, *projb, *n2w, *n2b, *fc1w, *fc1b, *fc2w, *fc2b;
    // decoder only
    const float *qw, *qb, *kvw, *kvb, *cprojw, *cprojb, *n3w, *n3b, *nyw, *nyb;
};

struct RcuW { const float *c1w, *c1b, *c2w, *c2b; };
struct FusionW { RcuW r1, r2; const float *ow, *ob; };
struct HeadW {
    const float *a0w, *a0b, *a0tw, *a0tb, *a1w, *a1b, *a1tw, *a1tb, *a2w, *a2b, *a3w, *a3b, *a3cw, *a3cb;
    const float* rn[4];
    FusionW ref[4];   // ref[0] = refinenet1 ... ref[3] = refinenet4
    const float *h0w, *h0b, *h2w, *h2b, *h4w, *h4b;
};

}  // namespace a3r
using namespace a3r;

struct a3r_model_s {
    a3r_model_config cfg;
    std::map<std::string, WRef> w;
    bool finalized = false;
    std::vector<BlockW> enc, pc, dec1, dec2;
    HeadW head[2];
    const float *pe_w, *pe_b, *pepc_w, *pepc_b, *encn_w, *encn_b, *de_w, *de_b, *decn_w, *decn_b;
    std::vector<const float*> zc_w, zc_b;
    const float *rope_cos = nullptr, *rope_sin = nullptr;
    std::vector<float> host_cos, host_sin;
    std::map<std::string, std::pair<const float*, size_t>> taps;
    // nn.Linear weights also kept in bf3 form (gemm_bf3.hip) unless A3R_GEMM=f32: fp32 pointer -> bf3 twin in `packed`
    bool use_bf3 = true;
    int fh2_passes = 3;       // fh2 matrix passes per product: 3 = fp32-grade; A3R_GEMM=f16 -> 1 (plain fp16 operands, reduced precision)
    int products = 6;         // bf3 plane products per multiply: 6 = fp32-accurate; A3R_GEMM=bf3x3 -> 3, A3R_GEMM=bf16 -> 1 (reduced precision)
    std::map<const float*, const void*> w3;
    // transformer nn.Linear weights in fh2 form (two fp16 planes, gemm_fh2.hip) unless A3R_GEMM names another mode:
    // fp32 pointer -> (fh2 twin in `packed`, the power-of-two scale it was stored with)
    bool use_fh2 = true;
    bool conv_fh2 = true;     // fh2 mode: the DPT maps / 3x3 convs on the fh2 kernel too (A3R_CONV=bf3 keeps them on the three-plane bf16 kernel)
    std::map<const float*, std::pair<const void*, float>> w2;
    static constexpr int MAX_POS = 256;
    // fh2 range control (fh2.h, RANGE; a3r_model_range_check): one power-of-two scale per fh2-producing site of the launch plan, kept
    // per plan phase (forward / encode / decode walk different plans), and the device words the sites record max |stored value| in
    static constexpr int MAX_SITES = 1024;
    std::vector<float> site_scale[3];
    unsigned* stats = nullptr;       // [MAX_SITES], inside the packed buffer
    int last_phase = -1, last_sites = 0;
    // LayerNorm -> fh2 sites need no run-time statistics: |y| <= max|gamma| sqrt(D) + max|beta| whatever the input, so their scale is
    // fixed at finalize from the two weight vectors (gamma pointer -> scale); see ln_static_scale
    std::map<const float*, float> ln_scale;
    int tap_level = 0;               // a3r_model_set_tap_level: decoder level copied to the taps "level" / "pc0" (parity tests), 0 = none
};

static int n_pc_blocks(const a3r_model_config& c) { return c.dec_depth / 2 - 2; }

extern "C" int a3r_model_create(const a3r_model_config* cfg, a3r_model_t* out) {
    A3R_CHECK_ARG(cfg && out, "a3r_model_create: null argument");
    A3R_CHECK_ARG(cfg->enc_embed_dim == cfg->enc_num_heads * 64 && cfg->dec_embed_dim == cfg->dec_num_heads * 64,
                  "a3r_model_create: head_dim must be 64 (enc %d/%d, dec %d/%d)", cfg->enc_embed_dim, cfg->enc_num_heads,
                  cfg->dec_embed_dim, cfg->dec_num_heads);
    A3R_CHECK_ARG(cfg->patch_size == 16, "a3r_model_create: patch_size must be 16");
    A3R_CHECK_ARG(cfg->dec_depth > 9, "a3r_model_create: dec_depth must be > 9 (dpt_head.py:101)");
    A3R_CHECK_ARG(cfg->enc_embed_dim % 32 == 0 && cfg->dec_embed_dim % 32 == 0 && cfg->feature_dim % 32 == 0 &&
                      cfg->last_dim % 32 == 0, "a3r_model_create: widths must be multiples of 32");
    for (int i = 0; i < 4; i++) A3R_CHECK_ARG(cfg->layer_dims[i] % 32 == 0, "a3r_model_create: layer_dims must be multiples of 32");
    a3r_model_s* m = new (std::nothrow) a3r_model_s();
    A3R_CHECK_ARG(m, "out of host memory");
    m->cfg = *cfg;
    if (const char* e = getenv("A3R_GEMM")) {
        const std::string mode(e);
        // f32: exact-fp32 MFMA kernels; bf3: every GEMM on the exact three-plane bf16 form (6 passes); bf3x3 / bf16: reduced-precision
        // bf3 modes; anything else (default, "fh2"): transformer GEMMs on the two-plane fp16 form (3 passes), DPT convs / attention on bf3
        // f16: the fh2 kernels with ONE pass per product (plain fp16 operands under the same range control): the fast 16-bit mode
        m->use_bf3 = mode != "f32";
        m->products = mode == "bf16" ? 1 : mode == "bf3x3" ? 3 : 6;
        m->use_fh2 = !(mode == "f32" || mode == "bf3" || mode == "bf3x3" || mode == "bf16");
        m->fh2_passes = mode == "f16" ? 1 : 3;
    }
    if (const char* e = getenv("A3R_CONV")) m->conv_fh2 = std::string(e) != "bf3";
    m->conv_fh2 = m->conv_fh2 && m->use_bf3 && m->use_fh2;
    // sized here so that the host-side sizing pass (a3r_model_workspace_bytes) works before finalize
    m->enc.assign(cfg->enc_depth, BlockW());
    m->pc.assign(n_pc_blocks(*cfg), BlockW());
    m->dec1.assign(cfg->dec_depth, BlockW());
    m->dec2.assign(cfg->dec_depth, BlockW());
    m->zc_w.assign(n_pc_blocks(*cfg) + 1, nullptr);
    m->zc_b.assign(n_pc_blocks(*cfg) + 1, nullptr);
    *out = m;
    return A3R_OK;
}

extern "C" int a3r_model_destroy(a3r_model_t m) {
    delete m;
    return A3R_OK;
}

extern "C" int a3r_model_set_weight(a3r_model_t m, const char* name, const float* ptr, int ndim, const int64_t* shape) {
    A3R_CHECK_ARG(m && name && ptr && ndim >= 1 && ndim <= 4 && shape, "a3r_model_set_weight: bad argument");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "a3r_model_set_weight: %s is not 16-byte aligned", name);
    WRef r;
    r.p = ptr;
    r.shape.assign(shape, shape + ndim);
    m->w[name] = r;
    m->finalized = false;
    return A3R_OK;
}

// ------------------------------------------------------------------------------------------- packing plan
namespace {
// kind 0: conv3x3 [Cout=a,Cin=b]; 1: convT [Cin=a,Cout=b,s]; 2: kv-concat (D=a); 3: rope tables; 4: bf3 twin of the [a, b] nn.Linear weight `name`
struct PackItem { std::string name; int kind; int a, b, s; size_t off; };

std::vector<PackItem> pack_plan(a3r_model_s* m, size_t* total) {
    const a3r_model_config& c = m->cfg;
    std::vector<PackItem> v;
    size_t off = 0;
    auto add = [&](const std::string& n, int kind, int a, int b, int s, size_t nfloat) {
        v.push_back({n, kind, a, b, s, off});
        off = align_up(off + nfloat * 4, 256);
    };
    const int F = c.feature_dim;
    for (int h = 1; h <= 2; h++) {
        const std::string p = "downstream_head" + std::to_string(h) + ".dpt.";
        for (int i = 0; i < 4; i++) add(p + "scratch.layer" + std::to_string(i + 1) + "_rn.weight", 0, F, c.layer_dims[i], 0, (size_t)F * c.layer_dims[i] * 9);
        for (int r = 1; r <= 4; r++)
            for (int u = 1; u <= 2; u++)
                for (int k = 1; k <= 2; k++)
                    add(p + "scratch.refinenet" + std::to_string(r) + ".resConfUnit" + std::to_string(u) + ".conv" + std::to_string(k) + ".weight", 0, F, F, 0, (size_t)F * F * 9);
        add(p + "head.0.weight", 0, F / 2, F, 0, (size_t)(F / 2) * F * 9);
        add(p + "head.2.weight", 0, c.last_dim, F / 2, 0, (size_t)c.last_dim * (F / 2) * 9);
        add(p + "act_postprocess.3.1.weight", 0, c.layer_dims[3], c.layer_dims[3], 0, (size_t)c.layer_dims[3] * c.layer_dims[3] * 9);
        add(p + "act_postprocess.0.1.weight", 1, c.layer_dims[0], c.layer_dims[0], 4, (size_t)c.layer_dims[0] * c.layer_dims[0] * 16);
        add(p + "act_postprocess.1.1.weight", 1, c.layer_dims[1], c.layer_dims[1], 2, (size_t)c.layer_dims[1] * c.layer_dims[1] * 4);
    }
    const int D = c.dec_embed_dim;
    for (int d = 0; d < 2; d++)
        for (int i = 0; i < c.dec_depth; i++) {
            const std::string p = std::string(d ? "dec_blocks2." : "dec_blocks.") + std::to_string(i) + ".cross_attn.";
            add(p + "kv", 2, D, 0, 0, (size_t)2 * D * D + 2 * D);
        }
    v.push_back({"rope", 3, 0, 0, 0, off});
    off = align_up(off + (size_t)2 * a3r_model_s::MAX_POS * 16 * 4, 256);
    v.push_back({"range_stats", 6, 0, 0, 0, off});        // kind 6: the range statistics words of the fh2 sites
    off = align_up(off + (size_t)a3r_model_s::MAX_SITES * 4, 256);
    if (m->use_bf3) {
        auto twin = [&](const std::string& n, int N, int K) {
            v.push_back({n, 4, N, K, 0, off});
            off = align_up(off + a3r_bf3_w_bytes(N, K), 256);
        };
        // kind 5: fh2 twin (+ 256 bytes behind it: scratch of the max|w| reduction that fixes its scale)
        auto twin_lin = [&](const std::string& n, int N, int K) {
            if (!m->use_fh2) { twin(n, N, K); return; }
            v.push_back({n, 5, N, K, 0, off});
            off = align_up(off + a3r_fh2_bytes(N, K) + 256, 256);
        };
        auto block = [&](const std::string& p, int Dm, bool cross) {
            twin_lin(p + ".attn.qkv.weight", 3 * Dm, Dm); twin_lin(p + ".attn.proj.weight", Dm, Dm);
            twin_lin(p + ".mlp.fc1.weight", Dm * c.mlp_ratio, Dm); twin_lin(p + ".mlp.fc2.weight", Dm, Dm * c.mlp_ratio);
            if (cross) {
                twin_lin(p + ".cross_attn.projq.weight", Dm, Dm); twin_lin(p + ".cross_attn.kv", 2 * Dm, Dm);
                twin_lin(p + ".cross_attn.proj.weight", Dm, Dm);
            }
        };
        const int E = c.enc_embed_dim;
        for (int i = 0; i < c.enc_depth; i++) block("enc_blocks." + std::to_string(i), E, false);
        for (int i = 0; i < n_pc_blocks(c); i++) block("dec_blocks_pc." + std::to_string(i), D, false);
        for (int i = 0; i < c.dec_depth; i++) {
            block("dec_blocks." + std::to_string(i), D, true);
            block("dec_blocks2." + std::to_string(i), D, true);
        }
        twin_lin("patch_embed.proj.weight", E, 768);
        twin_lin("patch_embed_point_cloud.proj.weight", D, 768);
        twin_lin("decoder_embed.weight", D, E);
        for (int i = 0; i <= n_pc_blocks(c); i++) twin_lin("zero_convs." + std::to_string(i) + ".0.weight", D, D);
        if (m->use_fh2)        // the 1x1 adapters of the DPT heads (act_postprocess.*.0): [layer_dim, D or E, 1, 1] = an nn.Linear weight
            for (int h = 1; h <= 2; h++) {
                const std::string p = "downstream_head" + std::to_string(h) + ".dpt.act_postprocess.";
                twin_lin(p + "0.0.weight", c.layer_dims[0], E);
                for (int i = 1; i < 4; i++) twin_lin(p + std::to_string(i) + ".0.weight", c.layer_dims[i], D);
            }
        // DPT heads: every packed 3x3 conv weight [Cout, 9 Cin] and the 1x1 out_conv of the fusion blocks
        std::vector<PackItem> convs;
        for (const PackItem& it : v)
            if (it.kind == 0) convs.push_back(it);
        auto twin_conv = [&](const std::string& n, int N, int K) {
            if (m->conv_fh2) twin_lin(n, N, K);
            else twin(n, N, K);
        };
        for (const PackItem& it : convs) twin_conv(it.name, it.a, 9 * it.b);
        for (int h = 1; h <= 2; h++)
            for (int r = 1; r <= 4; r++)
                twin_conv("downstream_head" + std::to_string(h) + ".dpt.scratch.refinenet" + std::to_string(r) + ".out_conv.weight", F, F);
    }
    *total = off;
    return v;
}
}  // namespace

extern "C" size_t a3r_model_packed_bytes(a3r_model_t m) {
    if (!m) return 0;
    size_t total = 0;
    pack_plan(m, &total);
    return total;
}

static int need(a3r_model_s* m, const std::string& name, std::vector<int64_t> shape, const float** out) {
    auto it = m->w.find(name);
    if (it == m->w.end()) {
        set_error("a3r_model_finalize: missing weight '%s'", name.c_str());
        return A3R_ESTATE;
    }
    if (it->second.shape != shape) {
        std::string got, want;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        for (auto d : shape) want += std::to_string(d) + ",";
        set_error("a3r_model_finalize: weight '%s' has shape [%s] but [%s] is required", name.c_str(), got.c_str(), want.c_str());
        return A3R_EINVAL;
    }
    *out = it->second.p;
    return A3R_OK;
}

#define NEED(name, out, ...)                                         \
    do {                                                             \
        if (int rc__ = need(m, name, {__VA_ARGS__}, out)) return rc__; \
    } while (0)

static int bind_block(a3r_model_s* m, const std::string& p, int D, int hidden, bool cross, BlockW* b) {
    NEED(p + ".norm1.weight", &b->n1w, D); NEED(p + ".norm1.bias", &b->n1b, D);
    NEED(p + ".attn.qkv.weight", &b->qkvw, 3 * D, D); NEED(p + ".attn.qkv.bias", &b->qkvb, 3 * D);
    NEED(p + ".attn.proj.weight", &b->projw, D, D); NEED(p + ".attn.proj.bias", &b->projb, D);
    NEED(p + ".norm2.weight", &b->n2w, D); NEED(p + ".norm2.bias", &b->n2b, D);
    NEED(p + ".mlp.fc1.weight", &b->fc1w, hidden, D); NEED(p + ".mlp.fc1.bias", &b->fc1b, hidden);
    NEED(p + ".mlp.fc2.weight", &b->fc2w, D, hidden); NEED(p + ".mlp.fc2.bias", &b->fc2b, D);
    if (cross) {
        const float* t;
        NEED(p + ".cross_attn.projq.weight", &b->qw, D, D); NEED(p + ".cross_attn.projq.bias", &b->qb, D);
        NEED(p + ".cross_attn.projk.weight", &t, D, D); NEED(p + ".cross_attn.projk.bias", &t, D);
        NEED(p + ".cross_attn.projv.weight", &t, D, D); NEED(p + ".cross_attn.projv.bias", &t, D);
        NEED(p + ".cross_attn.proj.weight", &b->cprojw, D, D); NEED(p + ".cross_attn.proj.bias", &b->cprojb, D);
        NEED(p + ".norm3.weight", &b->n3w, D); NEED(p + ".norm3.bias", &b->n3b, D);
        NEED(p + ".norm_y.weight", &b->nyw, D); NEED(p + ".norm_y.bias", &b->nyb, D);
    }
    return A3R_OK;
}

// Scale of a LayerNorm -> fh2 output from its weights alone.  Hard bound: |y| <= g sqrt(D) + b (g = max|gamma|, b = max|beta|; the
// normalised row has unit variance, so no entry exceeds sqrt(D - 1)).  1 unless that bound could pass 2^15 (then the power of two that
// keeps it below) or the typical magnitude g + b is under 2^-3 (then raised until g + b reaches [2^-1, 1), never past the bound).
static float ln_static_scale(float g, float b, int D) {
    const double bound = (double)g * std::sqrt((double)D) + (double)b, typ = (double)g + (double)b;
    if (!(bound > 0.0)) return 1.f;
    int e;
    std::frexp(bound, &e);                         // bound = f 2^e, f in [0.5, 1)
    const int kmax = 15 - e;                       // 2^kmax * bound < 2^15
    int k = 0;
    if (typ < 0.125) {
        int et;
        std::frexp(typ > 0.0 ? typ : bound, &et);
        k = -et;                                   // 2^k * typ in [0.5, 1)
    }
    if (k > kmax) k = kmax;
    k = k < -40 ? -40 : k > 40 ? 40 : k;
    return std::ldexp(1.f, k);
}

extern "C" int a3r_model_finalize(a3r_model_t m, void* packed, size_t packed_bytes, void* stream) {
    A3R_CHECK_ARG(m && packed, "a3r_model_finalize: null argument");
    size_t total = 0;
    std::vector<PackItem> plan = pack_plan(m, &total);
    A3R_CHECK_ARG(packed_bytes >= total, "a3r_model_finalize: packed buffer too small (%zu < %zu)", packed_bytes, total);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 255) == 0, "a3r_model_finalize: packed buffer must be 256-byte aligned");
    const a3r_model_config& c = m->cfg;
    const int E = c.enc_embed_dim, D = c.dec_embed_dim, F = c.feature_dim, L = c.last_dim;
    hipStream_t st = as_stream(stream);
    char* pk = static_cast<char*>(packed);
    std::map<std::string, const float*> packed_ptr;
    // --- repack
    for (const PackItem& it : plan) {
        float* dst = reinterpret_cast<float*>(pk + it.off);
        if (it.kind == 4 || it.kind == 5) continue;      // after binding (shapes are validated there)
        if (it.kind == 6) {
            m->stats = reinterpret_cast<unsigned*>(dst);
            A3R_HIP(hipMemsetAsync(dst, 0, (size_t)a3r_model_s::MAX_SITES * 4, st));
            continue;
        }
        if (it.kind == 0) {
            const float* src;
            NEED(it.name, &src, it.a, it.b, 3, 3);
            if (int rc = a3r_pack_conv3x3(src, dst, it.a, it.b, stream)) return rc;
        } else if (it.kind == 1) {
            const float* src;
            NEED(it.name, &src, it.a, it.b, it.s, it.s);
            if (int rc = a3r_pack_convT(src, dst, it.a, it.b, it.s, stream)) return rc;
        } else if (it.kind == 2) {
            const std::string p = it.name.substr(0, it.name.size() - 2);   // strip "kv"
            const float *kw, *kb, *vw, *vb;
            NEED(p + "projk.weight", &kw, D, D); NEED(p + "projk.bias", &kb, D);
            NEED(p + "projv.weight", &vw, D, D); NEED(p + "projv.bias", &vb, D);
            const size_t DD = (size_t)D * D * 4;
            A3R_HIP(hipMemcpyAsync(dst, kw, DD, hipMemcpyDeviceToDevice, st));
            A3R_HIP(hipMemcpyAsync(dst + (size_t)D * D, vw, DD, hipMemcpyDeviceToDevice, st));
            A3R_HIP(hipMemcpyAsync(dst + (size_t)2 * D * D, kb, D * 4, hipMemcpyDeviceToDevice, st));
            A3R_HIP(hipMemcpyAsync(dst + (size_t)2 * D * D + D, vb, D * 4, hipMemcpyDeviceToDevice, st));
        } else {
            m->host_cos.resize(a3r_model_s::MAX_POS * 16);
            m->host_sin.resize(a3r_model_s::MAX_POS * 16);
            a3r_rope_table_host(m->host_cos.data(), m->host_sin.data(), a3r_model_s::MAX_POS, c.rope_base);
            A3R_HIP(hipMemcpyAsync(dst, m->host_cos.data(), m->host_cos.size() * 4, hipMemcpyHostToDevice, st));
            A3R_HIP(hipMemcpyAsync(dst + a3r_model_s::MAX_POS * 16, m->host_sin.data(), m->host_sin.size() * 4, hipMemcpyHostToDevice, st));
            m->rope_cos = dst;
            m->rope_sin = dst + a3r_model_s::MAX_POS * 16;
        }
        packed_ptr[it.name] = dst;
    }
    // --- bind
    NEED("patch_embed.proj.weight", &m->pe_w, E, 3, 16, 16); NEED("patch_embed.proj.bias", &m->pe_b, E);
    NEED("patch_embed_point_cloud.proj.weight", &m->pepc_w, D, 3, 16, 16); NEED("patch_embed_point_cloud.proj.bias", &m->pepc_b, D);
    NEED("enc_norm.weight", &m->encn_w, E); NEED("enc_norm.bias", &m->encn_b, E);
    NEED("decoder_embed.weight", &m->de_w, D, E); NEED("decoder_embed.bias", &m->de_b, D);
    NEED("dec_norm.weight", &m->decn_w, D); NEED("dec_norm.bias", &m->decn_b, D);
    m->enc.assign(c.enc_depth, BlockW());
    for (int i = 0; i < c.enc_depth; i++)
        if (int rc = bind_block(m, "enc_blocks." + std::to_string(i), E, E * c.mlp_ratio, false, &m->enc[i])) return rc;
    const int npc = n_pc_blocks(c);
    m->pc.assign(npc, BlockW());
    for (int i = 0; i < npc; i++)
        if (int rc = bind_block(m, "dec_blocks_pc." + std::to_string(i), D, D * c.mlp_ratio, false, &m->pc[i])) return rc;
    m->dec1.assign(c.dec_depth, BlockW());
    m->dec2.assign(c.dec_depth, BlockW());
    for (int i = 0; i < c.dec_depth; i++) {
        if (int rc = bind_block(m, "dec_blocks." + std::to_string(i), D, D * c.mlp_ratio, true, &m->dec1[i])) return rc;
        if (int rc = bind_block(m, "dec_blocks2." + std::to_string(i), D, D * c.mlp_ratio, true, &m->dec2[i])) return rc;
        const float* kv1 = packed_ptr["dec_blocks." + std::to_string(i) + ".cross_attn.kv"];
        const float* kv2 = packed_ptr["dec_blocks2." + std::to_string(i) + ".cross_attn.kv"];
        m->dec1[i].kvw = kv1; m->dec1[i].kvb = kv1 + (size_t)2 * D * D;
        m->dec2[i].kvw = kv2; m->dec2[i].kvb = kv2 + (size_t)2 * D * D;
    }
    m->zc_w.assign(npc + 1, nullptr);
    m->zc_b.assign(npc + 1, nullptr);
    for (int i = 0; i <= npc; i++) {
        NEED("zero_convs." + std::to_string(i) + ".0.weight", &m->zc_w[i], D, D, 1);
        NEED("zero_convs." + std::to_string(i) + ".0.bias", &m->zc_b[i], D);
    }
    for (int h = 0; h < 2; h++) {
        HeadW& H = m->head[h];
        const std::string p = "downstream_head" + std::to_string(h + 1) + ".dpt.";
        const int* ld = c.layer_dims;
        NEED(p + "act_postprocess.0.0.weight", &H.a0w, ld[0], E, 1, 1); NEED(p + "act_postprocess.0.0.bias", &H.a0b, ld[0]);
        NEED(p + "act_postprocess.0.1.bias", &H.a0tb, ld[0]);
        NEED(p + "act_postprocess.1.0.weight", &H.a1w, ld[1], D, 1, 1); NEED(p + "act_postprocess.1.0.bias", &H.a1b, ld[1]);
        NEED(p + "act_postprocess.1.1.bias", &H.a1tb, ld[1]);
        NEED(p + "act_postprocess.2.0.weight", &H.a2w, ld[2], D, 1, 1); NEED(p + "act_postprocess.2.0.bias", &H.a2b, ld[2]);
        NEED(p + "act_postprocess.3.0.weight", &H.a3w, ld[3], D, 1, 1); NEED(p + "act_postprocess.3.0.bias", &H.a3b, ld[3]);
        NEED(p + "act_postprocess.3.1.bias", &H.a3cb, ld[3]);
        H.a0tw = packed_ptr[p + "act_postprocess.0.1.weight"];
        H.a1tw = packed_ptr[p + "act_postprocess.1.1.weight"];
        H.a3cw = packed_ptr[p + "act_postprocess.3.1.weight"];
        for (int i = 0; i < 4; i++) H.rn[i] = packed_ptr[p + "scratch.layer" + std::to_string(i + 1) + "_rn.weight"];
        for (int r = 0; r < 4; r++) {
            const std::string q = p + "scratch.refinenet" + std::to_string(r + 1) + ".";
            NEED(q + "out_conv.weight", &H.ref[r].ow, F, F, 1, 1); NEED(q + "out_conv.bias", &H.ref[r].ob, F);
            RcuW* rr[2] = {&H.ref[r].r1, &H.ref[r].r2};
            for (int u = 0; u < 2; u++) {
                const std::string qq = q + "resConfUnit" + std::to_string(u + 1) + ".";
                rr[u]->c1w = packed_ptr[qq + "conv1.weight"]; rr[u]->c2w = packed_ptr[qq + "conv2.weight"];
                NEED(qq + "conv1.bias", &rr[u]->c1b, F); NEED(qq + "conv2.bias", &rr[u]->c2b, F);
            }
        }
        H.h0w = packed_ptr[p + "head.0.weight"]; NEED(p + "head.0.bias", &H.h0b, F / 2);
        H.h2w = packed_ptr[p + "head.2.weight"]; NEED(p + "head.2.bias", &H.h2b, L);
        NEED(p + "head.4.weight", &H.h4w, 4, L, 1, 1); NEED(p + "head.4.bias", &H.h4b, 4);
    }
    // --- static scales of the LayerNorm -> fh2 sites (fh2 mode)
    m->ln_scale.clear();
    if (m->use_bf3 && m->use_fh2) {
        std::vector<std::pair<const float*, const float*>> lns;      // (gamma, beta) of every LayerNorm whose output feeds a GEMM
        auto add_block = [&](const BlockW& b, bool cross) {
            lns.push_back({b.n1w, b.n1b}); lns.push_back({b.n2w, b.n2b});
            if (cross) { lns.push_back({b.n3w, b.n3b}); lns.push_back({b.nyw, b.nyb}); }
        };
        for (const BlockW& b : m->enc) add_block(b, false);
        for (const BlockW& b : m->pc) add_block(b, false);
        for (const BlockW& b : m->dec1) add_block(b, true);
        for (const BlockW& b : m->dec2) add_block(b, true);
        float* scratch = reinterpret_cast<float*>(m->stats);          // two words of the (still unused) statistics area
        for (size_t i = 0; i < lns.size(); i++) {
            const int Dn = i < 2 * m->enc.size() ? E : D;
            float gb[2] = {0.f, 0.f};
            if (int rc = a3r_absmax(lns[i].first, Dn, scratch, stream)) return rc;
            if (int rc = a3r_absmax(lns[i].second, Dn, scratch + 1, stream)) return rc;
            A3R_HIP(hipMemcpyAsync(gb, scratch, 8, hipMemcpyDeviceToHost, st));
            A3R_HIP(hipStreamSynchronize(st));
            if (!std::isfinite(gb[0]) || !std::isfinite(gb[1])) {
                set_error("a3r_model_finalize: a LayerNorm weight contains non-finite values");
                return A3R_EINVAL;
            }
            m->ln_scale[lns[i].first] = ln_static_scale(gb[0], gb[1], Dn);
        }
        A3R_HIP(hipMemsetAsync(m->stats, 0, 8, st));
    }
    // --- bf3 twins of the nn.Linear weights
    m->w3.clear();
    m->w2.clear();
    for (const PackItem& it : plan) {
        if (it.kind != 4 && it.kind != 5) continue;
        const float* src = nullptr;
        auto pit = packed_ptr.find(it.name);
        if (pit != packed_ptr.end()) src = pit->second;                 // the concatenated cross-attention k/v projection
        else src = m->w.at(it.name).p;                                   // bound (and shape-checked) above
        void* dst = pk + it.off;
        if (it.kind == 4) {
            if (int rc = a3r_split_bf3_w(src, it.b, dst, it.a, it.b, stream)) return rc;     // weights: row-pair layout (bf3.h)
            m->w3[src] = dst;
        } else {
            // fh2 twin: scale = the power of two that puts max|w| into [2^12, 2^13) (one small reduction + read-back per matrix)
            float* scratch = reinterpret_cast<float*>(pk + it.off + a3r_fh2_bytes(it.a, it.b));
            if (int rc = a3r_absmax(src, (long)it.a * it.b, scratch, stream)) return rc;
            float amax = 0.f;
            A3R_HIP(hipMemcpyAsync(&amax, scratch, 4, hipMemcpyDeviceToHost, st));
            A3R_HIP(hipStreamSynchronize(st));
            if (!std::isfinite(amax)) {
                set_error("a3r_model_finalize: weight '%s' contains non-finite values", it.name.c_str());
                return A3R_EINVAL;
            }
            const float scale = a3r_fh2_weight_scale(amax);
            if (int rc = a3r_split_fh2(src, it.b, dst, it.a, it.b, scale, nullptr, stream)) return rc;
            m->w2[src] = {dst, scale};
        }
    }
    m->finalized = true;
    return A3R_OK;
}

// ------------------------------------------------------------------------------------------- launch plan
namespace {

struct Plan {
    a3r_model_s* m;
    Arena ar;
    void* stream;
    int rc = A3R_OK;
    bool trace = getenv("A3R_TRACE") != nullptr;   // debugging aid: name + synchronise every op
    int opno = 0;
    bool dry() const { return ar.dry; }
    // ---- fh2 range control: every op that WRITES an fh2 tensor is a site (numbered in plan order) with its own power-of-two
    // scale; the scale travels with the buffer pointer to the ops that read it
    int phase = 0, site_no = 0;
    std::map<const void*, float> buf_scale;
    struct Site { float scale; unsigned* stat; };
    Site site(const void* buf, const void* alias = nullptr) {
        const int id = site_no++;
        if (dry() || rc != A3R_OK) return {1.f, nullptr};
        if (id >= a3r_model_s::MAX_SITES) {
            set_error("a3r_model_forward: more than %d fh2 sites in the launch plan", a3r_model_s::MAX_SITES);
            rc = A3R_ESTATE;
            return {1.f, nullptr};
        }
        std::vector<float>& sc = m->site_scale[phase];
        if ((int)sc.size() <= id) sc.resize(id + 1, 1.f);
        buf_scale[buf] = sc[id];
        if (alias) buf_scale[alias] = sc[id];
        return {sc[id], m->stats + id};
    }
    float scale_of(const void* buf) {
        if (dry() || rc != A3R_OK) return 1.f;
        auto it = buf_scale.find(buf);
        if (it == buf_scale.end()) {
            set_error("a3r_model_forward: internal: an fh2 operand without a recorded scale");
            rc = A3R_ESTATE;
            return 1.f;
        }
        return it->second;
    }
    bool skip() {
        if (ar.overflow && !rc) {
            set_error("a3r_model_forward: internal workspace plan overflow (sizing pass and launch pass disagree)");
            rc = A3R_ESTATE;
        }
        return dry() || rc != A3R_OK;
    }
    void traced(const char* what, int a = 0, int b = 0, int c = 0) {
        if (!trace || dry()) return;
        hipError_t e = hipStreamSynchronize(as_stream(stream));
        fprintf(stderr, "[a3r trace] op %d before %s(%d,%d,%d): previous ops %s\n", opno++, what, a, b, c,
                e == hipSuccess ? "ok" : hipGetErrorString(e));
        fflush(stderr);
    }

    a3r_epilogue epi(int kind, const float* bias, const float* resid = nullptr, const float* resid2 = nullptr) {
        a3r_epilogue e = {};
        e.epi = kind; e.bias = bias; e.resid = resid; e.resid2 = resid2;
        return e;
    }
    // ---- "GEMM input" (gin) buffers: [rows, K] fp32 in f32 mode, the bf3 form of it (1.5x the bytes, bf3.h) otherwise
    bool bf3() const { return m->use_bf3; }
    bool fh2() const { return m->use_bf3 && m->use_fh2; }    // transformer GEMM operands in fh2 form (attention operands stay bf3)
    // bf3 mode: buffers that are only ever read as the A operand of a GEMM (LayerNorm / attention / fc1+GELU outputs, split
    // activations) are kept in the row-pair form of the layout (bf3.h) when the row counts of both views are even
    bool pair = false;
    // (an fh2 matrix has exactly the bytes of the fp32 one)
    float* gin_alloc(size_t rows, int K) { return ar.alloc(bf3() && !fh2() ? rows * K * 3 / 2 : rows * K); }
    float* gin_scratch(size_t rows, int K) { return ar.alloc(fh2() ? rows * K : bf3() ? rows * K * 3 / 2 : 0); }     // only needed for splitting
    template <class T> T* gin_at(T* base, size_t rows, int K) const { return base + (bf3() && !fh2() ? rows * K * 3 / 2 : rows * K); }
    // ---- attention operands (q / k / v written by the RoPE projections): fh2 in fh2 mode, bf3 in the bf3 modes
    float* att_alloc(size_t rows, int K) { return ar.alloc(bf3() && !fh2() ? rows * K * 3 / 2 : rows * K); }
    template <class T> T* att_at(T* base, size_t rows, int K) const { return base + (bf3() && !fh2() ? rows * K * 3 / 2 : rows * K); }
    bool cf2() const { return m->conv_fh2; }                  // DPT maps in fh2 form (else bf3 form in every bf3 / fh2 mode)
    // maps that are only ever conv inputs (DPT)
    float* map_alloc(size_t rows, int K) { return ar.alloc(bf3() && !cf2() ? rows * K * 3 / 2 : rows * K); }
    // a "bf3 output" request of the DPT plan means "the conv-input form": fh2 when the convs run on the fh2 kernel
    a3r_epilogue map_epi(const a3r_epilogue& e0) const {
        a3r_epilogue e = e0;
        if (cf2()) {
            if (e.out_bf3) { e.out_bf3 = 0; e.out_fh2 = 1; }
            if (e.aux_bf3) { e.aux_fh2 = e.aux_bf3; e.aux_bf3 = nullptr; }
        }
        return e;
    }
    // column `col` (a multiple of 8) of a gin row
    const float* gin_col(const float* base, int col) const {
        return bf3() && !fh2() ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)col * 6) : base + col;
    }
    // fp32 activation -> GEMM input: a split pass into `scratch` in bf3 mode, the array itself otherwise
    const float* gin_from(const float* x, float* scratch, long M, int K) {
        if (!bf3()) return x;
        if (skip()) return scratch;
        traced("split_bf3", (int)M, K);
        if (fh2()) {
            const Site s = site(scratch);
            rc = a3r_split_fh2(x, K, scratch, M, K, s.scale, s.stat, stream);
        } else rc = pair ? a3r_split_bf3_w(x, K, scratch, M, K, stream) : a3r_split_bf3(x, K, scratch, M, K, stream);
        return scratch;
    }
    const std::pair<const void*, float>* twin2(const float* w) {
        auto it = m->w2.find(w);
        if (it == m->w2.end()) {
            set_error("a3r_model_forward: nn.Linear weight without an fh2 twin");
            rc = A3R_ESTATE;
            return nullptr;
        }
        return &it->second;
    }
    const void* twin(const float* w) {
        auto it = m->w3.find(w);
        if (it == m->w3.end()) {
            set_error("a3r_model_forward: nn.Linear weight without a bf3 twin");
            rc = A3R_ESTATE;
            return nullptr;
        }
        return it->second;
    }
    // the range fields of an fh2 launch: the operand's scale, and a site for the fh2 output (y itself or the aux twin)
    void range_epi(a3r_epilogue& e, const void* x2, const void* y) {
        e.x_scale = scale_of(x2);
        if (e.out_fh2 || e.aux_fh2) {
            const Site s = site(e.out_fh2 ? y : e.aux_fh2);
            e.out_scale = s.scale; e.out_absmax = s.stat;
        }
    }
    // nn.Linear on a GEMM-input buffer (bf3 MFMA path unless A3R_GEMM=f32); plain_x: xg is a plain-rows bf3 matrix (DPT maps)
    void linear(const float* xg, int lda, const float* w, float* y, int ldc, int M, int N, int K, const a3r_epilogue& e0,
                bool plain_x = false) {
        if (skip()) return;
        traced("linear", M, N, K);
        if (fh2() && (!plain_x || cf2())) {
            const auto* w2 = twin2(w);
            a3r_epilogue e = plain_x ? map_epi(e0) : e0;
            range_epi(e, xg, y);
            if (w2 && !rc) rc = a3r_linear_fh2(xg, w2->first, w2->second, y, ldc, M, N, K, &e, stream);
        } else if (bf3()) {
            const void* w3 = twin(w);
            a3r_epilogue e = e0;
            e.x_pair = (pair && !plain_x) ? 1 : 0;
            if (w3) rc = a3r_linear_bf3(xg, w3, y, ldc, M, N, K, &e, stream);
        } else {
            const a3r_epilogue& e = e0;
            rc = a3r_linear(xg, lda, w, y, ldc, M, N, K, &e, stream);
        }
    }
    // nn.Linear / 1x1 conv on a plain fp32 activation (DPT adapters), exact-fp32 MFMA
    void linear_f32(const float* x, int lda, const float* w, float* y, int ldc, int M, int N, int K, const a3r_epilogue& e) {
        if (skip()) return;
        traced("linear_f32", M, N, K);
        rc = a3r_linear(x, lda, w, y, ldc, M, N, K, &e, stream);
    }
    // the same-shape projection of both decoders (dec_blocks[i] on view 1, dec_blocks2[i] on view 2) in one launch
    void linear2(const float* x0, const float* x1, int lda, const float* w0, const float* w1, const float* b0, const float* b1,
                 float* y0, float* y1, int ldc, int M, int N, int K, a3r_epilogue e, const float* r0 = nullptr,
                 const float* r1 = nullptr) {
        if (skip()) return;
        traced("linear2", M, N, K);
        if (fh2()) {
            const auto *w20 = twin2(w0), *w21 = twin2(w1);
            if (!w20 || !w21) return;
            // the two sides' outputs are ONE site (one scale: the attention kernel reads both sides in one launch)
            Site so = {1.f, nullptr};
            if (e.out_fh2) so = site(y0, y1);
            a3r_group_ptrs_fh2 g[2] = {{x0, w20->first, y0, b0, r0, nullptr, w20->second, scale_of(x0), so.scale, so.stat},
                                       {x1, w21->first, y1, b1, r1, nullptr, w21->second, scale_of(x1), so.scale, so.stat}};
            if (!rc) rc = a3r_linear_fh2_grouped(g, 2, ldc, M, N, K, &e, stream);
        } else if (bf3()) {
            const void *w30 = twin(w0), *w31 = twin(w1);
            if (!w30 || !w31) return;
            a3r_group_ptrs_bf3 g[2] = {{x0, w30, y0, b0, r0, nullptr}, {x1, w31, y1, b1, r1, nullptr}};
            e.x_pair = pair ? 1 : 0;
            rc = a3r_linear_bf3_grouped(g, 2, ldc, M, N, K, &e, stream);
        } else {
            a3r_group_ptrs g[2] = {{x0, w0, y0, b0, r0, nullptr}, {x1, w1, y1, b1, r1, nullptr}};
            rc = a3r_linear_grouped(g, 2, lda, ldc, M, N, K, &e, stream);
        }
    }
    // ---- bf3-mode helpers of the DPT heads
    float* alloc3(size_t rows, int K) { return ar.alloc(cf2() ? rows * K : rows * K * 3 / 2); }      // a conv-input [rows, K] buffer (bf3 or fh2 form)
    void split(const float* x, float* y3, long M, int K) {
        if (skip()) return;
        traced("split_map", (int)M, K);
        if (cf2()) {
            const Site s = site(y3);
            rc = a3r_split_fh2(x, K, y3, M, K, s.scale, s.stat, stream);
        } else rc = a3r_split_bf3(x, K, y3, M, K, stream);
    }
    void conv3(const float* x3, const float* wp, float* y, int B, int H, int W, int Cin, int Cout, int stride, const a3r_epilogue& e0) {
        if (skip()) return;
        traced("conv3x3_map", H, W, Cin);
        if (cf2()) {
            const auto* w2 = twin2(wp);
            a3r_epilogue e = map_epi(e0);
            range_epi(e, x3, y);
            if (w2 && !rc) rc = a3r_conv3x3_fh2(x3, w2->first, w2->second, y, B, H, W, Cin, Cout, stride, &e, stream);
            return;
        }
        const void* w3 = twin(wp);
        if (w3) rc = a3r_conv3x3_bf3(x3, w3, y, B, H, W, Cin, Cout, stride, &e0, stream);
    }
    void up3(const float* x, float* y3, int B, int H, int W, int C, int Hc, int Wc) {
        if (skip()) return;
        traced("upsample2x_map", H, W, C);
        if (cf2()) {
            const Site s = site(y3);
            rc = a3r_upsample2x_fh2(x, y3, B, H, W, C, Hc, Wc, s.scale, s.stat, stream);
        } else rc = a3r_upsample2x_bf3(x, y3, B, H, W, C, Hc, Wc, stream);
    }
    // LayerNorm whose output feeds a GEMM (written directly in bf3 form in bf3 mode)
    void ln(const float* x, const float* w, const float* b, float* yg, int M, int D) {
        if (skip()) return;
        traced("layernorm", M, D);
        if (fh2()) {
            // static scale (ln_static_scale): no statistics, the kernel keeps its one-row-per-wave grid
            auto it = m->ln_scale.find(w);
            const float sc = it == m->ln_scale.end() ? 1.f : it->second;
            buf_scale[yg] = sc;
            rc = a3r_layernorm_fh2(x, w, b, yg, M, D, 1e-6f, sc, nullptr, stream);
        } else rc = bf3() ? a3r_layernorm_bf3(x, w, b, yg, M, D, 1e-6f, pair, stream) : a3r_layernorm(x, w, b, yg, M, D, 1e-6f, stream);
    }
    void ln_f32(const float* x, const float* w, const float* b, float* y, int M, int D) {
        if (skip()) return;
        traced("layernorm_f32", M, D);
        rc = a3r_layernorm(x, w, b, y, M, D, 1e-6f, stream);
    }
    // q, k, v, o are gin buffers (bf3 mode: straight from / to the projection GEMMs, no fp32 round trip)
    // kv_buf: the buffer k and v are columns of when it is not q's (cross-attention); o_alias: the second side's rows of o when the
    // consumer addresses them by their own pointer (linear2)
    void attn(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, int B, int H, int Nq, int Nk,
              const float* kv_buf = nullptr, const float* o_alias = nullptr) {
        if (skip()) return;
        traced("attention", B, Nq, Nk);
        if (fh2()) {
            const float sq = scale_of(q), skv = kv_buf ? scale_of(kv_buf) : sq;
            const Site so = site(o, o_alias);
            const a3r_fh2_attn_range r = {sq, skv, skv, so.scale, so.stat};
            if (!rc) rc = a3r_attention_fh2(q, ldq, k, ldk, v, ldv, o, ldo, B, H, Nq, Nk, &r, stream);
            return;
        }
        rc = bf3() ? a3r_attention_bf3(q, ldq, k, ldk, v, ldv, o, ldo, B, H, Nq, Nk, pair, stream)
                   : a3r_attention(q, ldq, k, ldk, v, ldv, o, ldo, B, H, Nq, Nk, stream);
    }
    void conv(const float* x, const float* wp, float* y, int B, int H, int W, int Cin, int Cout, int stride, const a3r_epilogue& e) {
        if (skip()) return;
        traced("conv3x3", H, W, Cin);
        rc = a3r_conv3x3(x, wp, y, B, H, W, Cin, Cout, stride, &e, stream);
    }
    void up(const float* x, float* y, int B, int H, int W, int C, int Hc, int Wc) {
        if (skip()) return;
        traced("upsample2x", H, W, C);
        rc = a3r_upsample2x(x, y, B, H, W, C, Hc, Wc, stream);
    }
    // projection feeding attention: RoPE on the leading columns, output in gin form
    a3r_epilogue rope_epi(const float* bias, int rope_cols, int ntok, int gw) {
        a3r_epilogue e = epi(A3R_EPI_ROPE, bias);
        if (fh2()) e.out_fh2 = 1;
        else e.out_bf3 = bf3() ? 1 : 0;
        e.rope_cols = rope_cols; e.tokens_per_image = ntok; e.grid_w = gw;
        e.rope_cos = m->rope_cos; e.rope_sin = m->rope_sin;
        return e;
    }

    // Block.forward blocks.py:127-130 on x [M, D] in place (self-attention over images of ntok tokens)
    // xn [M, D], qkv [M, 3 D], att [M, D]: gin buffers
    void self_block(const BlockW& w, float* x, const float* resid_src, int M, int D, int H, int ntok, int gw, float* xn,
                    float* qkv, float* att) {
        // x_out = resid_src + attn(LN1(resid_src)); then MLP in place on x
        ln(resid_src, w.n1w, w.n1b, xn, M, D);
        linear(xn, D, w.qkvw, qkv, 3 * D, M, 3 * D, D, rope_epi(w.qkvb, 2 * D, ntok, gw));
        attn(qkv, 3 * D, gin_col(qkv, D), 3 * D, gin_col(qkv, 2 * D), 3 * D, att, D, M / ntok, H, ntok, ntok);
        linear(att, D, w.projw, x, D, M, D, D, epi(A3R_EPI_RESID, w.projb, resid_src));
    }
    // hid: gin [M, hidden] -- fc1's GELU epilogue writes the GEMM-input form directly (Mlp blocks.py:73-77)
    a3r_epilogue gin_epi(int kind, const float* bias) {
        a3r_epilogue e = epi(kind, bias);
        if (fh2()) { e.out_fh2 = 1; return e; }
        e.out_bf3 = bf3() ? 1 : 0;
        e.out_pair = (bf3() && pair) ? 1 : 0;
        return e;
    }
    void mlp(const BlockW& w, const float* nw, const float* nb, float* x, int M, int D, int hidden, float* xn, float* hid) {
        ln(x, nw, nb, xn, M, D);
        linear(xn, D, w.fc1w, hid, hidden, M, hidden, D, gin_epi(A3R_EPI_GELU, w.fc1b));
        linear(hid, hidden, w.fc2w, x, D, M, D, hidden, epi(A3R_EPI_RESID, w.fc2b, x));
    }
};

// rcu (dpt_block.py:120-142): out = conv2(relu(conv1(relu(x)))) + x (+ extra)
void rcu(Plan& P, const RcuW& w, const float* x, const float* extra, float* tmp, float* out, int B, int H, int W, int F) {
    a3r_epilogue e1 = P.epi(A3R_EPI_RELU, w.c1b);
    e1.relu_a = 1;
    P.conv(x, w.c1w, tmp, B, H, W, F, F, 1, e1);
    a3r_epilogue e2 = extra ? P.epi(A3R_EPI_RESID2, w.c2b, x, extra) : P.epi(A3R_EPI_RESID, w.c2b, x);
    P.conv(tmp, w.c2w, out, B, H, W, F, F, 1, e2);
}

// FeatureFusionBlock_custom.forward (dpt_block.py:186-218): returns [B, Hc, Wc, F] with (Hc,Wc) = crop of (2H,2W)
// (`two` says whether xs[1] exists: pointers are null during the dry sizing pass, so it cannot be inferred from x1)
float* fusion(Plan& P, const FusionW& w, const float* x0, const float* x1, bool two, int B, int H, int W, int F, int Hc, int Wc) {
    Arena& ar = P.ar;
    const size_t n = (size_t)B * H * W * F;
    float* tmp = ar.alloc(n);
    float* cur;
    if (two) {
        float* s = ar.alloc(n);
        rcu(P, w.r1, x1, x0, tmp, s, B, H, W, F);     // output + resConfUnit1(xs[1])
        cur = s;
    } else {
        cur = const_cast<float*>(x0);
    }
    float* o = ar.alloc(n);
    rcu(P, w.r2, cur, nullptr, tmp, o, B, H, W, F);
    float* u = ar.alloc((size_t)B * Hc * Wc * F);
    P.up(o, u, B, H, W, F, Hc, Wc);
    float* r = ar.alloc((size_t)B * Hc * Wc * F);
    P.linear_f32(u, F, w.ow, r, F, B * Hc * Wc, F, F, P.epi(A3R_EPI_NONE, w.ob));
    return r;
}

// ---- the same blocks on the bf3 kernels.  A conv input lives in bf3 form; where the fp32 value is also needed (skip
// connections) the producer writes both (aux_bf3), pre-activated when the consumer is an RCU (which starts with a ReLU).
// x: fp32 [B,H,W,F]; xr3: bf3 of relu(x); tmp3: bf3 scratch; out: fp32; out_r3: bf3 of relu(out) or null
// (aux_relu = false: out_r3 is the bf3 form of out itself, for a consumer that does not start with a ReLU)
void rcu_bf3(Plan& P, const RcuW& w, const float* x, const float* xr3, const float* extra, float* tmp3, float* out, float* out_r3,
             int B, int H, int W, int F, bool aux_relu = true) {
    a3r_epilogue e1 = P.epi(A3R_EPI_RELU, w.c1b);
    e1.out_bf3 = 1;
    P.conv3(xr3, w.c1w, tmp3, B, H, W, F, F, 1, e1);
    a3r_epilogue e2 = extra ? P.epi(A3R_EPI_RESID2, w.c2b, x, extra) : P.epi(A3R_EPI_RESID, w.c2b, x);
    if (out_r3) { e2.aux_bf3 = out_r3; e2.aux_relu = aux_relu ? 1 : 0; }
    P.conv3(tmp3, w.c2w, out, B, H, W, F, F, 1, e2);
}

// returns fp32 [B, Hc, Wc, F], or its bf3 form when `last` (refinenet1's output only feeds head.0's 3x3 conv)
float* fusion_bf3(Plan& P, const FusionW& w, const float* x0, const float* x0r3, const float* x1, const float* x1r3, bool two, int B,
                  int H, int W, int F, int Hc, int Wc, bool last) {
    Arena& ar = P.ar;
    const size_t px = (size_t)B * H * W, n = px * F;
    float* tmp3 = P.alloc3(px, F);
    const float *cur, *cur3;
    if (two) {
        float* s = ar.alloc(n);
        float* sr3 = P.alloc3(px, F);
        rcu_bf3(P, w.r1, x1, x1r3, x0, tmp3, s, sr3, B, H, W, F);     // output + resConfUnit1(xs[1])
        cur = s; cur3 = sr3;
    } else {
        cur = x0; cur3 = x0r3;
    }
    static const bool ref_order = getenv("A3R_DPT_REF_ORDER") != nullptr;    // A/B switch: up-sample first, as the reference does
    if (ref_order) {
        float* o = ar.alloc(n);
        rcu_bf3(P, w.r2, cur, cur3, nullptr, tmp3, o, nullptr, B, H, W, F);
        const size_t opx = (size_t)B * Hc * Wc;
        float* u3 = P.alloc3(opx, F);
        P.up3(o, u3, B, H, W, F, Hc, Wc);
        a3r_epilogue e = P.epi(A3R_EPI_NONE, w.ob);
        float* r;
        if (last) { r = P.alloc3(opx, F); e.out_bf3 = 1; }
        else r = ar.alloc(opx * F);
        P.linear(u3, F, w.ow, r, F, (int)opx, F, F, e, /*plain_x=*/true);
        return r;
    }
    // out_conv (1x1, dpt_block.py:216) is applied BEFORE the bilinear 2x instead of after it: both are linear maps over
    // different axes (channels / pixels) and the interpolation weights sum to one, so conv(up(x)) + b == up(conv(x) + b) up
    // to fp32 rounding -- a quarter of the 1x1 GEMM's rows, and the full-resolution map is written once instead of three times.
    float* o = ar.alloc(n);
    float* o3 = P.alloc3(px, F);
    rcu_bf3(P, w.r2, cur, cur3, nullptr, tmp3, o, o3, B, H, W, F, /*aux_relu=*/false);
    float* lo = ar.alloc(n);
    P.linear(o3, F, w.ow, lo, F, (int)px, F, F, P.epi(A3R_EPI_NONE, w.ob), /*plain_x=*/true);
    const size_t opx = (size_t)B * Hc * Wc;
    float* r;
    if (last) {
        r = P.alloc3(opx, F);
        P.up3(lo, r, B, H, W, F, Hc, Wc);
    } else {
        r = ar.alloc(opx * F);
        P.up(lo, r, B, H, W, F, Hc, Wc);
    }
    return r;
}

// phase: 0 = whole forward from images; 1 = encoder only (img1 [B,...] -> feat_out [B,N,E]);
//        2 = decoder + heads from cached encoder features (feat1_in / feat2_in [B,N,E]).
int run_plan(a3r_model_s* m, bool dry, const float* img1, const float* img2, const float* pd1, const float* pd2, int B,
             int H, int W, float* pts1, float* conf1, float* pts2, float* conf2, void* ws, size_t ws_bytes, void* stream,
             size_t* peak, int phase = 0, const float* feat1_in = nullptr, const float* feat2_in = nullptr,
             float* feat_out = nullptr) {
    const a3r_model_config& c = m->cfg;
    const int E = c.enc_embed_dim, D = c.dec_embed_dim, F = c.feature_dim, L = c.last_dim;
    const int nh = H / 16, nw = W / 16, N = nh * nw, BN = B * N, M2 = 2 * BN;
    struct ProductsGuard {      // the handle's arithmetic mode for the duration of this plan
        int prev; bool on;
        ProductsGuard(bool on_, int p) : prev(6), on(on_) { if (on) prev = a3r_bf3_set_products(p); }
        ~ProductsGuard() { if (on) a3r_bf3_set_products(prev); }
    } products_guard(!dry && m->use_bf3, m->products);
    struct PassesGuard {
        int prev; bool on;
        PassesGuard(bool on_, int p) : prev(3), on(on_) { if (on) prev = a3r_fh2_set_passes(p); }
        ~PassesGuard() { if (on) a3r_fh2_set_passes(prev); }
    } passes_guard(!dry && m->use_bf3 && m->use_fh2, m->fh2_passes);
    Plan P;
    P.m = m; P.stream = stream; P.phase = phase;
    if (!dry && m->use_bf3 && m->use_fh2) {
        if (hipMemsetAsync(m->stats, 0, (size_t)a3r_model_s::MAX_SITES * 4, as_stream(stream)) != hipSuccess) {
            set_error("a3r_model_forward: clearing the range statistics failed");
            return A3R_EHIP;
        }
        m->last_phase = phase;
        m->last_sites = 0;
    }
    struct SitesGuard {      // however the plan returns: how many sites it walked
        a3r_model_s* m; Plan* P; bool on;
        ~SitesGuard() { if (on) m->last_sites = P->site_no < a3r_model_s::MAX_SITES ? P->site_no : a3r_model_s::MAX_SITES; }
    } sites_guard{m, &P, !dry};
    static const bool plain_act = getenv("A3R_BF3_PLAIN_ACT") != nullptr;      // A/B switch: keep every activation in plain rows
    P.pair = m->use_bf3 && !m->use_fh2 && BN % 2 == 0 && !plain_act;
    P.ar = {static_cast<char*>(ws), 0, ws_bytes, dry, 0};
    Arena& ar = P.ar;
    if (phase == 1) {
        // ---------------- encoder only, B images (_encode_image model.py:151-163): per-frame features for caching.
        // The reference re-encodes a frame for every pair it appears in; the result does not depend on the pair.
        float* cols = ar.alloc((size_t)BN * 768);
        float* cols3 = P.gin_scratch(BN, 768);
        float* x = ar.alloc((size_t)BN * E);
        float* xn = P.gin_alloc(BN, E);
        float* qkv = P.att_alloc(BN, 3 * E);
        float* att = P.gin_alloc(BN, E);
        float* hid = P.gin_alloc(BN, E * c.mlp_ratio);
        if (!dry) {
            const long sb = 3L * H * W, sc = (long)H * W, sy = W, sx = 1;
            if ((P.rc = a3r_patchify(img1, cols, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
        }
        P.linear(P.gin_from(cols, cols3, BN, 768), 768, m->pe_w, x, E, BN, E, 768, P.epi(A3R_EPI_NONE, m->pe_b));
        for (int i = 0; i < c.enc_depth; i++) {
            P.self_block(m->enc[i], x, x, BN, E, c.enc_num_heads, N, nw, xn, qkv, att);
            P.mlp(m->enc[i], m->enc[i].n2w, m->enc[i].n2b, x, BN, E, E * c.mlp_ratio, xn, hid);
        }
        P.ln_f32(x, m->encn_w, m->encn_b, dry ? nullptr : feat_out, BN, E);
        if (peak) *peak = ar.peak;
        return P.rc;
    }
    // ---------------- persistent buffers
    float* feat = ar.alloc((size_t)M2 * E);       // enc_norm output = level 0
    float* pc = ar.alloc((size_t)M2 * D);
    float* fbuf[4];
    for (int i = 0; i < 4; i++) fbuf[i] = ar.alloc((size_t)M2 * D);   // ping, pong, hook A, hook B
    float* dec_last = ar.alloc((size_t)M2 * D);
    // debug taps (a3r_model_set_tap_level): copies of one decoder level and of the point-cloud tokens before their first block
    float* tap_lvl = m->tap_level > 0 ? ar.alloc((size_t)M2 * D) : nullptr;
    float* tap_pc0 = m->tap_level > 0 ? ar.alloc((size_t)M2 * D) : nullptr;
    auto tap_copy = [&](float* dst, const float* src) {
        if (dry || P.rc != A3R_OK || !dst) return;
        if (hipMemcpyAsync(dst, src, (size_t)M2 * D * 4, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) {
            set_error("a3r_model_forward: tap copy failed");
            P.rc = A3R_EHIP;
        }
    };
    const size_t mark = ar.off;
    // ---------------- encoder (model.py:151-163)
    if (phase == 2) {
        // cached per-frame features: gather the two views' rows into the [2*B*N, E] layout the decoder expects
        float* cols = ar.alloc((size_t)M2 * 768);
        float* cols3 = P.gin_scratch(M2, 768);
        if (!dry) {
            const size_t bytes = (size_t)BN * E * sizeof(float);
            hipError_t e1 = hipMemcpyAsync(feat, feat1_in, bytes, hipMemcpyDeviceToDevice, as_stream(stream));
            hipError_t e2 = hipMemcpyAsync(feat + (size_t)BN * E, feat2_in, bytes, hipMemcpyDeviceToDevice, as_stream(stream));
            if (e1 != hipSuccess || e2 != hipSuccess) { set_error("a3r_model_decode: feature copy failed"); return A3R_EHIP; }
            const long sb = 3L * H * W, sc = 1, sy = 3L * W, sx = 3;
            if ((P.rc = a3r_patchify(pd1, cols, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
            if ((P.rc = a3r_patchify(pd2, cols + (size_t)BN * 768, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
        }
        P.linear(P.gin_from(cols, cols3, M2, 768), 768, m->pepc_w, pc, D, M2, D, 768, P.epi(A3R_EPI_NONE, m->pepc_b));
    } else {
        float* cols = ar.alloc((size_t)M2 * 768);
        float* cols3 = P.gin_scratch(M2, 768);
        float* x = ar.alloc((size_t)M2 * E);
        float* xn = P.gin_alloc(M2, E);
        float* qkv = P.att_alloc(M2, 3 * E);
        float* att = P.gin_alloc(M2, E);
        float* hid = P.gin_alloc(M2, E * c.mlp_ratio);
        if (!dry) {
            const long sb = 3L * H * W, sc = (long)H * W, sy = W, sx = 1;
            if ((P.rc = a3r_patchify(img1, cols, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
            if ((P.rc = a3r_patchify(img2, cols + (size_t)BN * 768, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
        }
        P.linear(P.gin_from(cols, cols3, M2, 768), 768, m->pe_w, x, E, M2, E, 768, P.epi(A3R_EPI_NONE, m->pe_b));
        for (int i = 0; i < c.enc_depth; i++) {
            P.self_block(m->enc[i], x, x, M2, E, c.enc_num_heads, N, nw, xn, qkv, att);
            P.mlp(m->enc[i], m->enc[i].n2w, m->enc[i].n2b, x, M2, E, E * c.mlp_ratio, xn, hid);
        }
        P.ln_f32(x, m->encn_w, m->encn_b, feat, M2, E);
        // point-map patch embedding (model.py:244-248); pred_depth is [B,H,W,3]: channel stride 1
        if (!dry) {
            const long sb = 3L * H * W, sc = 1, sy = 3L * W, sx = 3;
            if ((P.rc = a3r_patchify(pd1, cols, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
            if ((P.rc = a3r_patchify(pd2, cols + (size_t)BN * 768, B, 3, H, W, sb, sc, sy, sx, stream))) return P.rc;
        }
        P.linear(P.gin_from(cols, cols3, M2, 768), 768, m->pepc_w, pc, D, M2, D, 768, P.epi(A3R_EPI_NONE, m->pepc_b));
    }
    ar.off = mark;
    tap_copy(tap_pc0, pc);
    // ---------------- decoder (model.py:201-233)
    const int hook_a = c.dec_depth * 2 / 4, hook_b = c.dec_depth * 3 / 4;   // levels 6 and 9 for depth 12
    const float *lvl_a = nullptr, *lvl_b = nullptr;
    {
        const int hidden = D * c.mlp_ratio;
        float* xn = P.gin_alloc(M2, D);
        float* yn = P.gin_alloc(M2, D);
        float* qkv = P.att_alloc(M2, 3 * D);
        float* qb = P.att_alloc(M2, D);
        float* kv = P.att_alloc(M2, 2 * D);
        float* att = P.gin_alloc(M2, D);
        float* hid = P.gin_alloc(M2, hidden);
        float* pc3 = P.gin_scratch(M2, D);
        float* cur = fbuf[0];
        {
            float* feat3 = P.gin_scratch(M2, E);
            P.linear(P.gin_from(feat, feat3, M2, E), E, m->de_w, cur, D, M2, D, E, P.epi(A3R_EPI_NONE, m->de_b));
        }
        P.linear(P.gin_from(pc, pc3, M2, D), D, m->zc_w[0], cur, D, M2, D, D, P.epi(A3R_EPI_RESID, m->zc_b[0], cur));
        int next_free = 1;
        const int npc = n_pc_blocks(c);
        for (int i = 0; i < c.dec_depth; i++) {
            const int level = i + 1;
            float* nxt;
            if (level == hook_a) nxt = fbuf[2];
            else if (level == hook_b) nxt = fbuf[3];
            else { nxt = (cur == fbuf[0]) ? fbuf[1] : fbuf[0]; }
            (void)next_free;
            {
                // both decoders advance together: side s reads x = cur[s], y = cur[1-s] (model.py:218-220) and
                // every same-shape projection of the two sides is one grouped launch.
                const BlockW& w0 = m->dec1[i];
                const BlockW& w1 = m->dec2[i];
                const size_t S = (size_t)BN * D;                   // side stride in a [2*BN, D] buffer
                const float *x0 = cur, *x1 = cur + S;
                float *o0 = nxt, *o1 = nxt + S;
                // x = x + attn(norm1(x))                                   blocks.py:187
                float* xn1 = P.gin_at(xn, BN, D);                  // side 1 of the gin buffers
                float* yn1 = P.gin_at(yn, BN, D);
                P.ln(x0, w0.n1w, w0.n1b, xn, BN, D);
                P.ln(x1, w1.n1w, w1.n1b, xn1, BN, D);
                float* att1 = P.gin_at(att, BN, D);
                P.linear2(xn, xn1, D, w0.qkvw, w1.qkvw, w0.qkvb, w1.qkvb, qkv, P.att_at(qkv, BN, 3 * D), 3 * D, BN, 3 * D, D,
                          P.rope_epi(nullptr, 2 * D, N, nw));
                P.attn(qkv, 3 * D, P.gin_col(qkv, D), 3 * D, P.gin_col(qkv, 2 * D), 3 * D, att, D, 2 * B, c.dec_num_heads, N, N, nullptr, att1);
                P.linear2(att, att1, D, w0.projw, w1.projw, w0.projb, w1.projb, o0, o1, D, BN, D, D,
                          P.epi(A3R_EPI_RESID, nullptr), x0, x1);
                // y_ = norm_y(y); x = x + cross_attn(norm2(x), y_, y_)     blocks.py:188-189
                P.ln(x1, w0.nyw, w0.nyb, yn, BN, D);               // side 0 attends to view 2's tokens
                P.ln(x0, w1.nyw, w1.nyb, yn1, BN, D);
                P.ln(o0, w0.n2w, w0.n2b, xn, BN, D);
                P.ln(o1, w1.n2w, w1.n2b, xn1, BN, D);
                P.linear2(xn, xn1, D, w0.qw, w1.qw, w0.qb, w1.qb, qb, P.att_at(qb, BN, D), D, BN, D, D, P.rope_epi(nullptr, D, N, nw));
                P.linear2(yn, yn1, D, w0.kvw, w1.kvw, w0.kvb, w1.kvb, kv, P.att_at(kv, BN, 2 * D), 2 * D, BN, 2 * D, D,
                          P.rope_epi(nullptr, D, N, nw));
                P.attn(qb, D, kv, 2 * D, P.gin_col(kv, D), 2 * D, att, D, 2 * B, c.dec_num_heads, N, N, kv, att1);
                P.linear2(att, att1, D, w0.cprojw, w1.cprojw, w0.cprojb, w1.cprojb, o0, o1, D, BN, D, D,
                          P.epi(A3R_EPI_RESID, nullptr), o0, o1);
                // x = x + mlp(norm3(x))                                    blocks.py:190
                P.ln(o0, w0.n3w, w0.n3b, xn, BN, D);
                P.ln(o1, w1.n3w, w1.n3b, xn1, BN, D);
                float* hid1 = P.gin_at(hid, BN, hidden);
                P.linear2(xn, xn1, D, w0.fc1w, w1.fc1w, w0.fc1b, w1.fc1b, hid, hid1, hidden, BN, hidden, D,
                          P.gin_epi(A3R_EPI_GELU, nullptr));
                P.linear2(hid, hid1, hidden, w0.fc2w, w1.fc2w, w0.fc2b, w1.fc2b, o0, o1, D, BN, D, hidden,
                          P.epi(A3R_EPI_RESID, nullptr), o0, o1);
            }
            if (i < npc) {   // model.py:223-226
                P.self_block(m->pc[i], pc, pc, M2, D, c.dec_num_heads, N, nw, xn, qkv, att);
                P.mlp(m->pc[i], m->pc[i].n2w, m->pc[i].n2b, pc, M2, D, hidden, xn, hid);
                P.linear(P.gin_from(pc, pc3, M2, D), D, m->zc_w[i + 1], nxt, D, M2, D, D, P.epi(A3R_EPI_RESID, m->zc_b[i + 1], nxt));
            }
            if (level == hook_a) lvl_a = nxt;
            if (level == hook_b) lvl_b = nxt;
            if (level == m->tap_level) tap_copy(tap_lvl, nxt);
            cur = nxt;
        }
        P.ln_f32(cur, m->decn_w, m->decn_b, dec_last, M2, D);   // model.py:231-232
    }
    ar.off = mark;
    if (!dry) {
        m->taps.clear();
        m->taps["feat"] = {feat, (size_t)M2 * E};
        m->taps["hook_a"] = {lvl_a, (size_t)M2 * D};
        m->taps["hook_b"] = {lvl_b, (size_t)M2 * D};
        m->taps["dec_last"] = {dec_last, (size_t)M2 * D};
        if (tap_lvl) { m->taps["level"] = {tap_lvl, (size_t)M2 * D}; m->taps["pc0"] = {tap_pc0, (size_t)M2 * D}; }
    }
    // ---------------- DPT heads (dpt_head.py:34-66), one per view, fp32
    for (int s = 0; s < 2; s++) {
        ar.off = mark;
        const HeadW& Hd = m->head[s];
        const int* ld = c.layer_dims;
        const float* t0 = feat + (size_t)s * BN * E;
        const float* t1 = lvl_a + (dry ? 0 : (size_t)s * BN * D);
        const float* t2 = lvl_b + (dry ? 0 : (size_t)s * BN * D);
        const float* t3 = dec_last + (size_t)s * BN * D;
        const int h3 = (nh + 2 - 3) / 2 + 1, w3 = (nw + 2 - 3) / 2 + 1;
        // act_postprocess (dpt_block.py:353-405)
        // 1x1 adapters: in fh2 mode a split pass + the fh2 GEMM (5x the exact-fp32 MFMA kernel's rate) instead of a3r_linear
        auto adapter = [&](const float* t, int K, const float* w, const float* b, float* y, int N) {
            if (!P.fh2()) { P.linear_f32(t, K, w, y, N, BN, N, K, P.epi(A3R_EPI_NONE, b)); return; }
            const size_t keep = ar.off;
            float* t2 = P.gin_scratch(BN, K);
            P.linear(P.gin_from(t, t2, BN, K), K, w, y, N, BN, N, K, P.epi(A3R_EPI_NONE, b));
            ar.off = keep;                                          // the split is dead once the GEMM is enqueued (stream order)
        };
        float* a0 = ar.alloc((size_t)BN * ld[0]);
        adapter(t0, E, Hd.a0w, Hd.a0b, a0, ld[0]);
        float* l0 = ar.alloc((size_t)BN * 16 * ld[0]);
        {
            a3r_epilogue e = P.epi(A3R_EPI_PIXSHUF, Hd.a0tb);
            e.ps_s = 4; e.ps_h = nh; e.ps_w = nw; e.ps_cout = ld[0];
            P.linear_f32(a0, ld[0], Hd.a0tw, l0, ld[0], BN, 16 * ld[0], ld[0], e);
        }
        float* a1 = ar.alloc((size_t)BN * ld[1]);
        adapter(t1, D, Hd.a1w, Hd.a1b, a1, ld[1]);
        float* l1 = ar.alloc((size_t)BN * 4 * ld[1]);
        {
            a3r_epilogue e = P.epi(A3R_EPI_PIXSHUF, Hd.a1tb);
            e.ps_s = 2; e.ps_h = nh; e.ps_w = nw; e.ps_cout = ld[1];
            P.linear_f32(a1, ld[1], Hd.a1tw, l1, ld[1], BN, 4 * ld[1], ld[1], e);
        }
        float* l2 = ar.alloc((size_t)BN * ld[2]);
        adapter(t2, D, Hd.a2w, Hd.a2b, l2, ld[2]);
        float* a3 = ar.alloc((size_t)BN * ld[3]);
        adapter(t3, D, Hd.a3w, Hd.a3b, a3, ld[3]);
        float* l3 = P.map_alloc((size_t)B * h3 * w3, ld[3]);          // only feeds layer4_rn's conv: bf3 in bf3 mode
        if (!P.bf3()) {
            P.conv(a3, Hd.a3cw, l3, B, nh, nw, ld[3], ld[3], 2, P.epi(A3R_EPI_NONE, Hd.a3cb));
        } else {
            float* a33 = P.alloc3(BN, ld[3]);
            P.split(a3, a33, BN, ld[3]);
            a3r_epilogue e = P.epi(A3R_EPI_NONE, Hd.a3cb);
            e.out_bf3 = 1;
            P.conv3(a33, Hd.a3cw, l3, B, nh, nw, ld[3], ld[3], 2, e);
        }
        float* h2 = nullptr;
        bool fused_tail = false;
        const int Hh = 8 * nh, Wh = 8 * nw;
        if (!P.bf3()) {
            // scratch.layer_rn (no bias)
            float* r0 = ar.alloc((size_t)BN * 16 * F);
            P.conv(l0, Hd.rn[0], r0, B, 4 * nh, 4 * nw, ld[0], F, 1, P.epi(A3R_EPI_NONE, nullptr));
            float* r1 = ar.alloc((size_t)BN * 4 * F);
            P.conv(l1, Hd.rn[1], r1, B, 2 * nh, 2 * nw, ld[1], F, 1, P.epi(A3R_EPI_NONE, nullptr));
            float* r2 = ar.alloc((size_t)BN * F);
            P.conv(l2, Hd.rn[2], r2, B, nh, nw, ld[2], F, 1, P.epi(A3R_EPI_NONE, nullptr));
            float* r3 = ar.alloc((size_t)B * h3 * w3 * F);
            P.conv(l3, Hd.rn[3], r3, B, h3, w3, ld[3], F, 1, P.epi(A3R_EPI_NONE, nullptr));
            // refinement (dpt_head.py:57-60)
            float* p4 = fusion(P, Hd.ref[3], r3, nullptr, false, B, h3, w3, F, nh, nw);
            float* p3 = fusion(P, Hd.ref[2], p4, r2, true, B, nh, nw, F, 2 * nh, 2 * nw);
            float* p2 = fusion(P, Hd.ref[1], p3, r1, true, B, 2 * nh, 2 * nw, F, 4 * nh, 4 * nw);
            float* p1 = fusion(P, Hd.ref[0], p2, r0, true, B, 4 * nh, 4 * nw, F, 8 * nh, 8 * nw);
            // head (dpt_block.py:323-330)
            float* h0 = ar.alloc((size_t)B * Hh * Wh * (F / 2));
            P.conv(p1, Hd.h0w, h0, B, Hh, Wh, F, F / 2, 1, P.epi(A3R_EPI_NONE, Hd.h0b));
            float* hu = ar.alloc((size_t)B * H * W * (F / 2));
            P.up(h0, hu, B, Hh, Wh, F / 2, H, W);
            h2 = ar.alloc((size_t)B * H * W * L);
            P.conv(hu, Hd.h2w, h2, B, H, W, F / 2, L, 1, P.epi(A3R_EPI_RELU, Hd.h2b));

        } else {
            // conv inputs in bf3 form (small maps: plain split passes)
            float* l03 = P.alloc3((size_t)BN * 16, ld[0]);
            P.split(l0, l03, (long)BN * 16, ld[0]);
            float* l13 = P.alloc3((size_t)BN * 4, ld[1]);
            P.split(l1, l13, (long)BN * 4, ld[1]);
            float* l23 = P.alloc3(BN, ld[2]);
            P.split(l2, l23, BN, ld[2]);
            // scratch.layer_rn (no bias): fp32 for the skip connection + pre-activated bf3 for the first RCU conv
            float* r0 = ar.alloc((size_t)BN * 16 * F);
            float* r0r3 = P.alloc3((size_t)BN * 16, F);
            float* r1 = ar.alloc((size_t)BN * 4 * F);
            float* r1r3 = P.alloc3((size_t)BN * 4, F);
            float* r2 = ar.alloc((size_t)BN * F);
            float* r2r3 = P.alloc3(BN, F);
            float* r3 = ar.alloc((size_t)B * h3 * w3 * F);
            float* r3r3 = P.alloc3((size_t)B * h3 * w3, F);
            auto rn_epi = [&](float* aux) { a3r_epilogue e = P.epi(A3R_EPI_NONE, nullptr); e.aux_bf3 = aux; e.aux_relu = 1; return e; };
            P.conv3(l03, Hd.rn[0], r0, B, 4 * nh, 4 * nw, ld[0], F, 1, rn_epi(r0r3));
            P.conv3(l13, Hd.rn[1], r1, B, 2 * nh, 2 * nw, ld[1], F, 1, rn_epi(r1r3));
            P.conv3(l23, Hd.rn[2], r2, B, nh, nw, ld[2], F, 1, rn_epi(r2r3));
            P.conv3(l3, Hd.rn[3], r3, B, h3, w3, ld[3], F, 1, rn_epi(r3r3));      // l3 is bf3 in this mode
            // refinement (dpt_head.py:57-60)
            float* p4 = fusion_bf3(P, Hd.ref[3], r3, r3r3, nullptr, nullptr, false, B, h3, w3, F, nh, nw, false);
            float* p3 = fusion_bf3(P, Hd.ref[2], p4, nullptr, r2, r2r3, true, B, nh, nw, F, 2 * nh, 2 * nw, false);
            float* p2 = fusion_bf3(P, Hd.ref[1], p3, nullptr, r1, r1r3, true, B, 2 * nh, 2 * nw, F, 4 * nh, 4 * nw, false);
            float* p13 = fusion_bf3(P, Hd.ref[0], p2, nullptr, r0, r0r3, true, B, 4 * nh, 4 * nw, F, 8 * nh, 8 * nw, true);
            // head (dpt_block.py:323-330)
            float* h0 = ar.alloc((size_t)B * Hh * Wh * (F / 2));
            P.conv3(p13, Hd.h0w, h0, B, Hh, Wh, F, F / 2, 1, P.epi(A3R_EPI_NONE, Hd.h0b));
            float* hu3 = P.alloc3((size_t)B * H * W, F / 2);
            P.up3(h0, hu3, B, Hh, Wh, F / 2, H, W);
            if (P.cf2() && L == 128) {
                // head.2 (3x3 conv + ReLU), head.4 (1x1 conv 128 -> 4) and the postprocess in ONE launch: the [B H W, 128] map is
                // neither written nor read back (8.4 GB per head and step at 42 pairs), dpt_block.py:323-329 + postprocess.py:10-58
                a3r_epilogue e = P.epi(A3R_EPI_HEAD, Hd.h2b);
                e.head_w = Hd.h4w; e.head_b = Hd.h4b; e.head_conf = s ? conf2 : conf1;
                P.conv3(hu3, Hd.h2w, s ? pts2 : pts1, B, H, W, F / 2, L, 1, e);
                fused_tail = true;
            } else {
                h2 = ar.alloc((size_t)B * H * W * L);
                P.conv3(hu3, Hd.h2w, h2, B, H, W, F / 2, L, 1, P.epi(A3R_EPI_RELU, Hd.h2b));
            }
        }
        if (!fused_tail && !P.skip())
            P.rc = a3r_head_final(h2, Hd.h4w, Hd.h4b, s ? pts2 : pts1, s ? conf2 : conf1, (long)B * H * W, L, stream);
    }
    if (peak) *peak = ar.peak;
    return P.rc;
}
}  // namespace

extern "C" size_t a3r_model_workspace_bytes(a3r_model_t m, int B, int H, int W) {
    if (!m || B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) return 0;
    size_t peak = 0;
    run_plan(m, true, nullptr, nullptr, nullptr, nullptr, B, H, W, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &peak);
    return peak;
}

extern "C" int a3r_model_forward(a3r_model_t m, const float* img1, const float* img2, const float* pd1, const float* pd2,
                                 int B, int H, int W, float* pts1, float* conf1, float* pts2, float* conf2, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    A3R_CHECK_ARG(m, "a3r_model_forward: null handle");
    if (!m->finalized) {
        set_error("a3r_model_forward: a3r_model_finalize has not been called");
        return A3R_ESTATE;
    }
    A3R_CHECK_ARG(img1 && img2 && pd1 && pd2 && pts1 && conf1 && pts2 && conf2 && workspace, "a3r_model_forward: null pointer");
    A3R_CHECK_ARG(B > 0, "a3r_model_forward: batch must be positive");
    A3R_CHECK_ARG(H > 0 && H % 16 == 0, "Input image height (%d) is not a multiple of patch size (16).", H);
    A3R_CHECK_ARG(W > 0 && W % 16 == 0, "Input image width (%d) is not a multiple of patch size (16).", W);
    A3R_CHECK_ARG(H / 16 < a3r_model_s::MAX_POS && W / 16 < a3r_model_s::MAX_POS, "a3r_model_forward: image too large for the RoPE table");
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "a3r_model_forward: workspace must be 256-byte aligned");
    const size_t need_bytes = a3r_model_workspace_bytes(m, B, H, W);
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_model_forward: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return run_plan(m, false, img1, img2, pd1, pd2, B, H, W, pts1, conf1, pts2, conf2, workspace, workspace_bytes, stream, nullptr);
}

static int check_forward_args(a3r_model_t m, int B, int H, int W, const void* workspace, const char* who) {
    A3R_CHECK_ARG(m, "%s: null handle", who);
    if (!m->finalized) {
        set_error("%s: a3r_model_finalize has not been called", who);
        return A3R_ESTATE;
    }
    A3R_CHECK_ARG(workspace, "%s: null workspace", who);
    A3R_CHECK_ARG(B > 0, "%s: batch must be positive", who);
    A3R_CHECK_ARG(H > 0 && H % 16 == 0, "Input image height (%d) is not a multiple of patch size (16).", H);
    A3R_CHECK_ARG(W > 0 && W % 16 == 0, "Input image width (%d) is not a multiple of patch size (16).", W);
    A3R_CHECK_ARG(H / 16 < a3r_model_s::MAX_POS && W / 16 < a3r_model_s::MAX_POS, "%s: image too large for the RoPE table", who);
    A3R_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "%s: workspace must be 256-byte aligned", who);
    return A3R_OK;
}

extern "C" size_t a3r_model_encode_workspace_bytes(a3r_model_t m, int B, int H, int W) {
    if (!m || B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) return 0;
    size_t peak = 0;
    run_plan(m, true, nullptr, nullptr, nullptr, nullptr, B, H, W, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &peak, 1);
    return peak;
}

extern "C" int a3r_model_encode(a3r_model_t m, const float* img, int B, int H, int W, float* feat_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (int rc = check_forward_args(m, B, H, W, workspace, "a3r_model_encode")) return rc;
    A3R_CHECK_ARG(img && feat_out, "a3r_model_encode: null pointer");
    const size_t need_bytes = a3r_model_encode_workspace_bytes(m, B, H, W);
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_model_encode: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return run_plan(m, false, img, nullptr, nullptr, nullptr, B, H, W, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                    stream, nullptr, 1, nullptr, nullptr, feat_out);
}

extern "C" int a3r_model_decode(a3r_model_t m, const float* feat1, const float* feat2, const float* pd1, const float* pd2, int B,
                                int H, int W, float* pts1, float* conf1, float* pts2, float* conf2, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (int rc = check_forward_args(m, B, H, W, workspace, "a3r_model_decode")) return rc;
    A3R_CHECK_ARG(feat1 && feat2 && pd1 && pd2 && pts1 && conf1 && pts2 && conf2, "a3r_model_decode: null pointer");
    const size_t need_bytes = a3r_model_workspace_bytes(m, B, H, W);      // the whole-forward plan bounds the decode plan
    A3R_CHECK_ARG(workspace_bytes >= need_bytes, "a3r_model_decode: workspace too small (%zu < %zu)", workspace_bytes, need_bytes);
    return run_plan(m, false, nullptr, nullptr, pd1, pd2, B, H, W, pts1, conf1, pts2, conf2, workspace, workspace_bytes, stream,
                    nullptr, 2, feat1, feat2, nullptr);
}

extern "C" int a3r_model_set_tap_level(a3r_model_t m, int level) {
    A3R_CHECK_ARG(m && level >= 0 && level <= m->cfg.dec_depth, "a3r_model_set_tap_level: level must be in 0..dec_depth");
    m->tap_level = level;
    return A3R_OK;
}

extern "C" int a3r_model_tap(a3r_model_t m, const char* name, const float** ptr, size_t* count) {
    A3R_CHECK_ARG(m && name && ptr && count, "a3r_model_tap: null argument");
    auto it = m->taps.find(name);
    A3R_CHECK_ARG(it != m->taps.end(), "a3r_model_tap: unknown tap '%s' (run a forward first)", name);
    *ptr = it->second.first;
    *count = it->second.second;
    return A3R_OK;
}

// ------------------------------------------------------------------------------------------- fh2 range control
extern "C" int a3r_model_range_check(a3r_model_t m, void* stream, int* n_adjusted, int* n_nonfinite) {
    A3R_CHECK_ARG(m && n_adjusted && n_nonfinite, "a3r_model_range_check: null argument");
    *n_adjusted = 0;
    *n_nonfinite = 0;
    if (!(m->use_bf3 && m->use_fh2) || m->last_phase < 0 || m->last_sites <= 0) return A3R_OK;     // fp32-range modes / nothing ran
    std::vector<unsigned> st((size_t)m->last_sites);
    A3R_HIP(hipMemcpyAsync(st.data(), m->stats, st.size() * 4, hipMemcpyDeviceToHost, as_stream(stream)));
    A3R_HIP(hipStreamSynchronize(as_stream(stream)));
    std::vector<float>& sc = m->site_scale[m->last_phase];
    if (sc.size() < st.size()) sc.resize(st.size(), 1.f);
    for (size_t i = 0; i < st.size(); i++) {
        float stored;                                        // max |scale * x| the site wrote
        static_assert(sizeof(float) == sizeof(unsigned), "bit pattern");
        memcpy(&stored, &st[i], 4);
        if (st[i] == 0) continue;                            // an all-zero tensor (or a site that did not run) says nothing
        if (!std::isfinite(stored)) { (*n_nonfinite)++; continue; }
        if (stored >= 0.25f && stored <= 32768.f) continue;  // [2^-2, 2^15]: fp32-grade (fh2.h)
        // new scale: the power of two that puts max |x| = stored / scale into [2^11, 2^12)
        int e;
        std::frexp(stored / sc[i], &e);                      // = f 2^e, f in [0.5, 1)   (the division by a power of two is exact)
        int k = 12 - e;
        k = k < -40 ? -40 : k > 40 ? 40 : k;              // (products of two scales stay far inside fp32)
        sc[i] = std::ldexp(1.f, k);
        (*n_adjusted)++;
    }
    return A3R_OK;
}

extern "C" int a3r_model_range_stats(a3r_model_t m, void* stream, float* stored_absmax, int capacity, int* n_sites) {
    A3R_CHECK_ARG(m && n_sites && (stored_absmax || capacity == 0), "a3r_model_range_stats: bad argument");
    *n_sites = m->last_sites;
    const int n = capacity < m->last_sites ? capacity : m->last_sites;
    if (n <= 0 || !m->stats) return A3R_OK;
    A3R_HIP(hipMemcpyAsync(stored_absmax, m->stats, (size_t)n * 4, hipMemcpyDeviceToHost, as_stream(stream)));
    A3R_HIP(hipStreamSynchronize(as_stream(stream)));
    return A3R_OK;
}

extern "C" int a3r_model_range_scales(a3r_model_t m, int phase, float* scales, int capacity, int* n_sites) {
    A3R_CHECK_ARG(m && n_sites && phase >= 0 && phase < 3 && (scales || capacity == 0), "a3r_model_range_scales: bad argument");
    const std::vector<float>& sc = m->site_scale[phase];
    *n_sites = (int)sc.size();
    for (int i = 0; i < capacity && i < (int)sc.size(); i++) scales[i] = sc[i];
    return A3R_OK;
}

extern "C" int a3r_model_reset_ranges(a3r_model_t m) {
    A3R_CHECK_ARG(m, "a3r_model_reset_ranges: null handle");
    for (auto& v : m->site_scale) v.clear();
    return A3R_OK;
}
