// VAE decoder engine: sequences the implicit-GEMM conv3d and elementwise kernels of one
// SimpleVideoDecoder pass (reference LTX_2_MLX/model/video_vae/simple_decoder.py:446-563) on a
// caller stream.  Activations are channels-last bf16 [T][H][W][C]; three ping-pong buffers.
#include <math.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ltx2hip.h"
#include <stdlib.h>

#include "gemm_epilogue.h"
#include "rowops.h"

#define TRY(expr)                       \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != LTX2_OK) return rc_; \
    } while (0)

namespace {
constexpr long SPLITK_WS_BYTES = 64L << 20;        // fp32 partial slabs of the small-M / long-K convs (gemm_v4.hip split-K)
struct Wt {
    const void* p;
    int dtype;
    long n;
};
inline long align_up(long v, long a = 256) { return (v + a - 1) / a * a; }
inline int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}
}  // namespace

struct ltx2_vae {
    ltx2_vae_config cfg{};
    float timestep_multiplier = 1000.f;
    std::unordered_map<std::string, Wt> weights;
    char* ws = nullptr;
    long ws_bytes = 0;
};

namespace {

const void* vfind(ltx2_vae* c, const std::string& name, int dtype, long numel) {
    auto it = c->weights.find(name);
    if (it == c->weights.end()) {
        ltx2_set_error("vae: missing weight '%s'", name.c_str());
        return nullptr;
    }
    if (it->second.dtype != dtype || it->second.n != numel) {
        ltx2_set_error("vae: weight '%s' has dtype %d / numel %ld, expected dtype %d / numel %ld", name.c_str(),
                       it->second.dtype, it->second.n, dtype, numel);
        return nullptr;
    }
    return it->second.p;
}
bool vhas(ltx2_vae* c, const std::string& name) { return c->weights.find(name) != c->weights.end(); }

// Largest activation (elements) over the whole pass + scratch for time embeddings.
struct Plan {
    long max_elems;
    int Tf, Hf, Wf, Cf;
};
Plan plan_sizes(const ltx2_vae_config& cfg, int T, int H, int W) {
    long ch = (long)cfg.base_channels * 8;
    long mx = (long)(T + 2) * (H + 2) * (W + 2) * (ch > cfg.latent_channels ? ch : cfg.latent_channels);
    for (int i = 0; i < cfg.n_blocks; ++i) {
        if (cfg.kind[i] == LTX2_VAE_UPSAMPLE) {
            const int ft = cfg.stride[i][0], fh = cfg.stride[i][1], fw = cfg.stride[i][2];
            T = T * ft - (ft > 1 ? 1 : 0);
            H *= fh;
            W *= fw;
            ch /= cfg.multiplier[i];
        }
        const long e = (long)(T + 2) * (H + 2) * (W + 2) * ch;      // the padded volume the v4 conv reads
        if (e > mx) mx = e;
    }
    return Plan{mx, T, H, W, (int)ch};
}

bool vae_v4_enabled() { return true; }     // (round 2's LTX2_VAE_V4=0 switch back to the tap-iterator kernels is gone)

GemmParams conv_params(const bf16* x, const bf16* w, const float* b, void* out, int T, int H, int W, int Cin, int Cout, int causal,
                       const bf16* res) {
    GemmParams p{};
    p.A = x;
    p.W = w;
    p.bias = b;
    p.out = out;
    p.M = T * H * W;
    p.N = Cout;
    p.K = 27 * Cin;
    p.taps_t = 3;
    p.ldo = Cout;
    p.res = res;
    p.ldres = Cout;
    p.T = T;
    p.H = H;
    p.Wd = W;
    p.Cin = Cin;
    p.cin_shift = ilog2(Cin);
    p.pad_front = causal ? 2 : 1;
    return p;
}

// res-block conv on the padded-volume kernel (gemm_v4.hip)?  The producer (pixel norm) must then write the padded layout.
bool conv_on_v4(int T, int H, int W, int Cin, int Cout, int epi, void* out) {
    if (!vae_v4_enabled()) return false;
    const GemmParams p = conv_params(nullptr, nullptr, nullptr, out, T, H, W, Cin, Cout, 0, epi == EPI_ADD_BF16 ? (const bf16*)out : nullptr);
    return gemm_v4_conv_supported(p, epi);
}

int conv(const bf16* x, const bf16* w, const float* b, void* out, int T, int H, int W, int Cin, int Cout, int causal,
         int epi, const bf16* res, int ft, int fh, int fw, int residual, hipStream_t st, int pad_zero = 0, int taps_t = 3) {
    GemmParams p{};
    p.A = x;
    p.W = w;
    p.bias = b;
    p.out = out;
    p.M = T * H * W;
    p.N = Cout;
    p.K = 9 * taps_t * Cin;
    p.taps_t = taps_t;
    p.pad_zero = pad_zero;
    p.ldo = Cout;
    p.res = res;
    p.ldres = Cout;
    p.T = T;
    p.H = H;
    p.Wd = W;
    p.Cin = Cin;
    p.cin_shift = ilog2(Cin);
    p.pad_front = taps_t == 1 ? 0 : (causal ? 2 : 1);
    if (epi == EPI_D2S_BF16) {
        const int sp = ft * fh * fw;
        LTX2_CHECK_ARG(Cout % sp == 0, "conv3d d2s: Cout=%d not divisible by stride product %d", Cout, sp);
        p.ft = ft;
        p.fh = fh;
        p.fw = fw;
        p.Cf = Cout / sp;
        LTX2_CHECK_ARG((p.Cf & (p.Cf - 1)) == 0, "conv3d d2s: Cf=%d must be a power of two", p.Cf);
        p.cf_shift = ilog2(p.Cf);
        p.drop_first = ft > 1 ? 1 : 0;
        p.d2s_residual = residual;
        p.c_d2s = Cin / sp;
        LTX2_CHECK_ARG(!residual || (Cin % sp == 0 && p.c_d2s > 0), "conv3d d2s: residual needs Cin %% stride product == 0");
    }
    return gemm_launch(p, epi, true, st);
}

}  // namespace

extern "C" {

int ltx2_conv3d_fused(const void* x, const void* w, const float* bias, void* out, int T, int H, int W, int Cin,
                      int Cout, int causal, int mode, const void* res, int ft, int fh, int fw, int residual,
                      int pad_zero, int kt, void* stream) {
    LTX2_CHECK_ARG(x && w && out, "conv3d: null operand");
    LTX2_CHECK_ARG(kt == 3 || kt == 1, "conv3d: temporal kernel size %d (3 or 1)", kt);
    LTX2_CHECK_ARG(pad_zero >= 0 && pad_zero <= 2 && !(pad_zero == 1 && causal), "conv3d: pad_zero in {0,1,2}; causal needs a replicate-T mode (0 or 2)");
    LTX2_CHECK_ARG(mode >= 0 && mode <= 2, "conv3d: mode %d", mode);
    LTX2_CHECK_ARG(mode != 1 || res, "conv3d: mode 1 needs a residual tensor");
    const int epi = mode == 0 ? EPI_BF16 : (mode == 1 ? EPI_ADD_BF16 : EPI_D2S_BF16);
    return conv((const bf16*)x, (const bf16*)w, bias, out, T, H, W, Cin, Cout, causal, epi, (const bf16*)res, ft, fh, fw,
                residual, (hipStream_t)stream, pad_zero, kt);
}

int ltx2_vae_create(const ltx2_vae_config* cfg, ltx2_vae** out) {
    LTX2_CHECK_ARG(cfg && out, "vae_create: null argument");
    LTX2_CHECK_ARG(cfg->n_blocks > 0 && cfg->n_blocks <= LTX2_VAE_MAX_BLOCKS, "vae_create: n_blocks=%d", cfg->n_blocks);
    LTX2_CHECK_ARG(cfg->latent_channels >= 64 && (cfg->latent_channels & (cfg->latent_channels - 1)) == 0,
                   "vae_create: latent_channels must be a power of two >= 64");
    long ch = (long)cfg->base_channels * 8;
    for (int i = 0; i < cfg->n_blocks; ++i) {
        if (cfg->kind[i] == LTX2_VAE_UPSAMPLE) {
            LTX2_CHECK_ARG(cfg->multiplier[i] >= 1, "vae_create: bad multiplier");
            ch /= cfg->multiplier[i];
        }
        LTX2_CHECK_ARG(ch >= 64 && (ch & (ch - 1)) == 0, "vae_create: channel width %ld must be a power of two >= 64", ch);
    }
    ltx2_vae* c = new ltx2_vae();
    c->cfg = *cfg;
    *out = c;
    return LTX2_OK;
}

void ltx2_vae_destroy(ltx2_vae* c) { delete c; }

int ltx2_vae_set_weight(ltx2_vae* c, const char* name, const void* ptr, int dtype, int64_t numel) {
    LTX2_CHECK_ARG(c && name && ptr, "vae_set_weight: null argument");
    c->weights[name] = Wt{ptr, dtype, (long)numel};
    return LTX2_OK;
}

int ltx2_vae_set_timestep_multiplier(ltx2_vae* c, float m) {
    LTX2_CHECK_ARG(c, "null context");
    c->timestep_multiplier = m;
    return LTX2_OK;
}

int ltx2_vae_out_frames(const ltx2_vae* c, int T) {
    if (!c) return -1;
    return plan_sizes(c->cfg, T, 1, 1).Tf;
}

int64_t ltx2_vae_workspace_bytes(const ltx2_vae* c, int T, int H, int W) {
    if (!c || T <= 0 || H <= 0 || W <= 0) return -1;
    const Plan p = plan_sizes(c->cfg, T, H, W);
    const long maxc = (long)c->cfg.base_channels * 8;
    return 3 * align_up(2 * p.max_elems) + align_up(4L * 256) + 2 * align_up(4L * 4 * maxc) + align_up(SPLITK_WS_BYTES) + 1024;
}

int ltx2_vae_bind_workspace(ltx2_vae* c, void* ptr, int64_t bytes) {
    LTX2_CHECK_ARG(c && ptr && bytes > 0, "vae_bind_workspace: bad argument");
    LTX2_CHECK_ARG(((uintptr_t)ptr & 255) == 0, "vae_bind_workspace: pointer must be 256-byte aligned");
    c->ws = (char*)ptr;
    c->ws_bytes = bytes;
    return LTX2_OK;
}

int ltx2_vae_decode(ltx2_vae* c, const float* latent, int T, int H, int W, float timestep, const float* noise,
                    int causal, float* video, void* stream) {
    LTX2_CHECK_ARG(c && latent && video, "vae_decode: null argument");
    LTX2_CHECK_ARG(H >= 2 && W >= 2 && T >= 1, "vae_decode: latent must be at least 1x2x2");
    const long need = ltx2_vae_workspace_bytes(c, T, H, W);
    if (!c->ws || c->ws_bytes < need) {
        ltx2_set_error("vae_decode: workspace too small (%ld bound, %ld needed)", c->ws_bytes, need);
        return LTX2_E_STATE;
    }
    hipStream_t st = (hipStream_t)stream;
    const ltx2_vae_config& cfg = c->cfg;
    const Plan pl = plan_sizes(cfg, T, H, W);
    const long bufsz = align_up(2 * pl.max_elems);
    bf16* X = (bf16*)c->ws;
    bf16* Y = (bf16*)(c->ws + bufsz);
    bf16* Z = (bf16*)(c->ws + 2 * bufsz);
    float* sinus = (float*)(c->ws + 3 * bufsz);
    const long maxc = (long)cfg.base_channels * 8;
    float* te_h = (float*)((char*)sinus + align_up(4L * 256));
    float* te = (float*)((char*)te_h + align_up(4L * 4 * maxc));
    void* skws = (char*)te + align_up(4L * 4 * maxc);
    const bool tcond = cfg.timestep_conditioning && timestep >= 0.f;
    const int CL = cfg.latent_channels;
    const float eps = 1e-6f;

#define W_BF16(name, n) (const bf16*)vfind(c, name, LTX2_DTYPE_BF16, (long)(n))
#define W_F32(name, n) (const float*)vfind(c, name, LTX2_DTYPE_F32, (long)(n))
#define NEED(ptr) \
    if (!(ptr)) return LTX2_E_STATE

    // denormalise (+ noise mix) and go channels-last (simple_decoder.py:492-498)
    const float* stdv = W_F32("vae.per_channel_statistics.std-of-means", CL);
    NEED(stdv);
    const float* meanv = W_F32("vae.per_channel_statistics.mean-of-means", CL);
    NEED(meanv);
    long P = (long)T * H * W;
    TRY(vae_prepare_latent_launch(latent, stdv, meanv, noise, tcond ? cfg.decode_noise_scale : 0.f, Y, CL, P, st));

    int ch = cfg.base_channels * 8;
    {
        const bf16* w = W_BF16("vae.decoder.conv_in.conv.weight", (long)ch * 27 * CL);
        NEED(w);
        const float* b = W_F32("vae.decoder.conv_in.conv.bias", ch);
        NEED(b);
        TRY(conv(Y, w, b, X, T, H, W, CL, ch, causal, EPI_BF16, nullptr, 1, 1, 1, 0, st));
    }
    if (tcond) TRY(timestep_sinusoid_launch(nullptr, 0, timestep, c->timestep_multiplier, 1, 256, sinus, nullptr, st));

    for (int i = 0; i < cfg.n_blocks; ++i) {
        const std::string pre = "vae.decoder.up_blocks." + std::to_string(i);
        if (cfg.kind[i] == LTX2_VAE_RES) {
            const float* tep = nullptr;
            const std::string tn = pre + ".time_embedder.timestep_embedder";
            if (tcond && vhas(c, tn + ".linear_1.weight")) {
                const long hid = c->weights[tn + ".linear_1.weight"].n / 256;
                const bf16* w1 = W_BF16(tn + ".linear_1.weight", hid * 256);
                NEED(w1);
                const float* b1 = W_F32(tn + ".linear_1.bias", hid);
                NEED(b1);
                const bf16* w2 = W_BF16(tn + ".linear_2.weight", 4L * ch * hid);
                NEED(w2);
                const float* b2 = W_F32(tn + ".linear_2.bias", 4L * ch);
                NEED(b2);
                LTX2_CHECK_ARG(hid <= 4 * maxc, "vae: time embedder hidden width too large");
                TRY(gemv_launch(sinus, 256, w1, b1, te_h, hid, 1, (int)hid, 256, 0, 1, st));
                TRY(gemv_launch(te_h, hid, w2, b2, te, 4L * ch, 1, 4 * ch, (int)hid, 0, 0, st));
                tep = te;
            }
            for (int j = 0; j < cfg.num_layers[i]; ++j) {
                const std::string rb = pre + ".res_blocks." + std::to_string(j);
                const float* tab = W_F32(rb + ".scale_shift_table", 4L * ch);
                NEED(tab);
                const bf16* w1 = W_BF16(rb + ".conv1.conv.weight", (long)ch * 27 * ch);
                NEED(w1);
                const float* b1 = W_F32(rb + ".conv1.conv.bias", ch);
                NEED(b1);
                const bf16* w2 = W_BF16(rb + ".conv2.conv.weight", (long)ch * 27 * ch);
                NEED(w2);
                const float* b2 = W_F32(rb + ".conv2.conv.bias", ch);
                NEED(b2);
                // rows: shift1, scale1, shift2, scale2 (simple_decoder.py:216-238)
                if (conv_on_v4(T, H, W, ch, ch, EPI_BF16, Z) && conv_on_v4(T, H, W, ch, ch, EPI_ADD_BF16, X)) {
                    // pixel norm writes the PADDED volume; the conv is a GEMM with one wave-uniform offset per tap
                    const int pf = causal ? 2 : 1;
                    TRY(pixnorm_mod_silu_padded_launch(X, Y, T, H, W, ch, eps, tab, tep, 0, 1, pf, st));
                    TRY(gemm_v4_conv_launch(conv_params(Y, w1, b1, Z, T, H, W, ch, ch, causal, nullptr), EPI_BF16, st, skws, SPLITK_WS_BYTES));
                    TRY(pixnorm_mod_silu_padded_launch(Z, Y, T, H, W, ch, eps, tab, tep, 2, 3, pf, st));
                    TRY(gemm_v4_conv_launch(conv_params(Y, w2, b2, X, T, H, W, ch, ch, causal, X), EPI_ADD_BF16, st, skws, SPLITK_WS_BYTES));
                } else {
                    TRY(pixnorm_mod_silu_launch(X, Y, P, ch, eps, tab, tep, 0, 1, st));
                    TRY(conv(Y, w1, b1, Z, T, H, W, ch, ch, causal, EPI_BF16, nullptr, 1, 1, 1, 0, st));
                    TRY(pixnorm_mod_silu_launch(Z, Y, P, ch, eps, tab, tep, 2, 3, st));
                    TRY(conv(Y, w2, b2, X, T, H, W, ch, ch, causal, EPI_ADD_BF16, X, 1, 1, 1, 0, st));
                }
            }
        } else {
            const int ft = cfg.stride[i][0], fh = cfg.stride[i][1], fw = cfg.stride[i][2];
            const int sp = ft * fh * fw;
            const int cout = sp * ch / cfg.multiplier[i];
            const bf16* w = W_BF16(pre + ".conv.conv.weight", (long)cout * 27 * ch);
            NEED(w);
            const float* b = W_F32(pre + ".conv.conv.bias", cout);
            NEED(b);
            GemmParams up = conv_params(Y, w, b, Z, T, H, W, ch, cout, causal, X);
            up.ft = ft;
            up.fh = fh;
            up.fw = fw;
            up.Cf = cout / sp;
            up.cf_shift = ilog2(up.Cf);
            up.drop_first = ft > 1 ? 1 : 0;
            up.d2s_residual = cfg.residual[i];
            up.c_d2s = ch / sp;
            if (vae_v4_enabled() && (up.Cf & (up.Cf - 1)) == 0 && (!cfg.residual[i] || (ch % sp == 0 && up.c_d2s > 0)) && gemm_v4_conv_supported(up, EPI_D2S_BF16)) {
                // the same padded-volume GEMM as the res-block convs: a copy pass writes the padding, the epilogue scatters depth to
                // space and adds the tiled d2s(x) residual read from the unpadded input
                TRY(pad_volume_launch(X, Y, T, H, W, ch, causal ? 2 : 1, st));
                TRY(gemm_v4_conv_launch(up, EPI_D2S_BF16, st));
            } else {
                TRY(conv(X, w, b, Z, T, H, W, ch, cout, causal, EPI_D2S_BF16, nullptr, ft, fh, fw, cfg.residual[i], st));
            }
            bf16* t = X;
            X = Z;
            Z = t;
            T = T * ft - (ft > 1 ? 1 : 0);
            H *= fh;
            W *= fw;
            ch /= cfg.multiplier[i];
            P = (long)T * H * W;
        }
    }

    // final norm / modulation / SiLU / conv_out / unpatchify (simple_decoder.py:528-553)
    const float* ltab = W_F32("vae.decoder.last_scale_shift_table", 2L * ch);
    NEED(ltab);
    const float* ltep = nullptr;
    const std::string ln = "vae.decoder.last_time_embedder.timestep_embedder";
    if (tcond && vhas(c, ln + ".linear_1.weight")) {
        const long hid = c->weights[ln + ".linear_1.weight"].n / 256;
        const bf16* w1 = W_BF16(ln + ".linear_1.weight", hid * 256);
        NEED(w1);
        const float* b1 = W_F32(ln + ".linear_1.bias", hid);
        NEED(b1);
        const bf16* w2 = W_BF16(ln + ".linear_2.weight", 2L * ch * hid);
        NEED(w2);
        const float* b2 = W_F32(ln + ".linear_2.bias", 2L * ch);
        NEED(b2);
        LTX2_CHECK_ARG(hid <= 4 * maxc, "vae: last time embedder hidden width too large");
        TRY(gemv_launch(sinus, 256, w1, b1, te_h, hid, 1, (int)hid, 256, 0, 1, st));
        TRY(gemv_launch(te_h, hid, w2, b2, te, 2L * ch, 1, 2 * ch, (int)hid, 0, 0, st));
        ltep = te;
    }
    {
        const bf16* w = W_BF16("vae.decoder.conv_out.conv.weight", 48L * 27 * ch);
        NEED(w);
        const float* b = W_F32("vae.decoder.conv_out.conv.bias", 48);
        NEED(b);
        if (conv_on_v4(T, H, W, ch, 48, EPI_BF16, Z)) {
            // round 4: conv_out on the asm-loop kernel too (layout 7: one 512 x 64 tile column, the 16 columns past Cout = 48 masked); the
            // pixel norm writes the padded volume as for the res-block convs
            TRY(pixnorm_mod_silu_padded_launch(X, Y, T, H, W, ch, eps, ltab, ltep, 0, 1, causal ? 2 : 1, st));
            TRY(gemm_v4_conv_launch(conv_params(Y, w, b, Z, T, H, W, ch, 48, causal, nullptr), EPI_BF16, st, skws, SPLITK_WS_BYTES));
        } else {
            TRY(pixnorm_mod_silu_launch(X, Y, P, ch, eps, ltab, ltep, 0, 1, st));
            TRY(conv(Y, w, b, Z, T, H, W, ch, 48, causal, EPI_BF16, nullptr, 1, 1, 1, 0, st));
        }
    }
    TRY(vae_unpatchify_launch(Z, video, T, H, W, st));
#undef W_BF16
#undef W_F32
#undef NEED
    return LTX2_OK;
}

}  // extern "C"
