// DiT engine: sequences the gfx950 kernels of one LTX-2 denoise step on a caller stream and owns
// the hipGraph of the distilled sampling loop.  Host-side only (no heavy kernels here).
//
// Restates the control flow of the reference's LTXModel.__call__ (LTX_2_MLX/model/transformer/
// model.py:776-881), BasicTransformerBlock.__call__ (transformer.py:191-238) and, for AudioVideo
// models, BasicAVTransformerBlock.__call__ (transformer.py:457-648) with
// MultiModalTransformerArgsPreprocessor (model.py:284-410).  Step-invariant work (caption
// projection, RoPE tables, and - when the text K/V are not sigma-modulated - the per-layer text
// cross-attention K/V) is hoisted into ltx2_dit_prepare*.
//
// A model is two "modalities" (video m[0], audio m[1]; audio absent for VideoOnly) that run the
// same per-block program (self-attention, text cross-attention, feed-forward) on their own widths,
// plus the audio<->video cross-modal attention between them.
#include <math.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ltx2hip.h"
#include "attention.h"
#include "gemm.h"
#include "rowops.h"

#define TRY(expr)                       \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != LTX2_OK) return rc_; \
    } while (0)

namespace {
struct Wt {
    const void* p;
    int dtype;
    long n;
};

struct AdaW {       // AdaLayerNormSingle (timestep_embedding.py:166-202)
    const bf16 *t1_w = nullptr, *t2_w = nullptr, *lin_w = nullptr;
    const float *t1_b = nullptr, *t2_b = nullptr, *lin_b = nullptr;
    int rows = 0;
};

struct AttnW {      // Attention (attention.py:144-253); self: fused qkv, cross: q + fused kv
    const bf16 *qkv_w = nullptr, *q_w = nullptr, *kv_w = nullptr, *o_w = nullptr, *g_w = nullptr;
    const float *qkv_b = nullptr, *q_b = nullptr, *kv_b = nullptr, *o_b = nullptr, *g_b = nullptr, *qn = nullptr,
                *kn = nullptr;
};

struct BlockW {     // one modality's share of a transformer block
    AttnW self, text;
    const bf16 *ff1_w = nullptr, *ff2_w = nullptr;
    const float *ff1_b = nullptr, *ff2_b = nullptr, *sst = nullptr, *prompt_sst = nullptr;
};

struct LayerW {
    BlockW m[2];
    AttnW a2v, v2a;
    const float* ca[2] = {nullptr, nullptr};     // scale_shift_table_a2v_ca_{video,audio}  [5][D]
};

struct ModW {       // per-modality model-level weights
    const bf16 *patch_w = nullptr, *cap1_w = nullptr, *cap2_w = nullptr, *proj_w = nullptr;
    const float *patch_b = nullptr, *cap1_b = nullptr, *cap2_b = nullptr, *proj_b = nullptr, *sst_out = nullptr;
    AdaW ada, prompt, cross_ss, cross_gate;
};

struct Mod {        // per-modality geometry + workspace
    int D = 0, H = 0, hd = 0, Cin = 0, Cout = 0;
    int N = 0, Npad = 0, S = 0, Spad = 0;
    float *x = nullptr, *sin_f = nullptr, *e1_f = nullptr, *e_f = nullptr, *emb = nullptr, *vel = nullptr, *x0 = nullptr,
          *cosb = nullptr, *sinb = nullptr, *ccos = nullptr, *csin = nullptr, *aux_e = nullptr, *prompt_emb = nullptr, *sst_all = nullptr, *comb = nullptr, *ca_all = nullptr, *ca_comb = nullptr,
          *cross_ss = nullptr, *cross_gate = nullptr, *glog = nullptr;
    int glog_parts = 1;                 // how the last gate_logits() call left m.glog: one array, or GATE_LOGIT_PARTS partial sums + glog_bias
    const float* glog_bias = nullptr;
    bf16 *lat = nullptr, *h = nullptr, *h2 = nullptr, *qkv = nullptr, *vt = nullptr, *att = nullptr, *ff = nullptr,
         *sin_b = nullptr, *e1_b = nullptr, *es_b = nullptr, *ctx_in = nullptr, *c1 = nullptr, *ctxp = nullptr,
         *ctxm = nullptr, *kv2 = nullptr, *vt2 = nullptr;
    const bf16* ctx = nullptr;      // projected text context (ctxp or ctx_in)
    unsigned long long* kmask = nullptr;   // text cross-attention key mask as 64-bit words (Modality.context_mask); used when has_kmask
    bool use_comb = false;             // this forward reads the combined AdaLN rows (comb) instead of (sst, emb)
    bool has_kmask = false;
    float* ts_tok = nullptr;        // per-token timesteps of a captured conditioned step: denoise_mask * sigma, formed on the device (per_token workspaces)
    float* qss = nullptr;           // text cross-attention with q_norm folded in: partial row sums of squares of the projected queries [N][D/64]
    float* knq = nullptr;           //   and k_norm.weight * q_norm.weight per layer [L][D] (the per-dim q weight moves onto the cached keys)
    bool qfold = false;             //   decided per prepare: the query projection runs on a kernel that writes the partial sums
    // round 6, the text cross-attention's pre-norm folded around the GEMMs (GemmParams::shadow / rf_parts; fold_supported()): h doubles as the bf16 shadow of the residual stream
    float* rss = nullptr;           //   partial sums of squares of the new residual rows, one per 256-column tile [D/256][rss_ld], left by attn1.to_out's epilogue
    long rss_ld = 0;                //   (the query projection turns them into its rows' RMS factors itself)
    int fold_max = 0;               //   what this geometry / model supports (prepare): 0 none, 1 the fold
    unsigned char* a8 = nullptr;    // fp8 compute: per-token e4m3fn codes of the current GEMM's activation operand [N][<= 4D]
    float* a8s = nullptr;           //              and their row scales [N]
};

inline long align_up(long v, long a = 256) { return (v + a - 1) / a * a; }
}  // namespace

struct ltx2_dit {
    ltx2_dit_config cfg{};
    bool av = false, v2 = false, gated = false;
    std::unordered_map<std::string, Wt> weights;
    std::vector<LayerW> layers;
    ModW mw[2];
    Mod m[2];
    bool resolved = false;
    char* ws = nullptr;
    long ws_bytes = 0;
    int per_token = 0;
    float* sigmas_dev = nullptr;
    // fp8-resident linear weights: the typed `const bf16*` fields of the weight structs then hold the CODES pointer; dense() looks it
    // up here to hand the GEMM (codes, per-row scale) instead of bf16 weights.  Per context (a second context over the same tensors
    // -- the video twin of an AudioVideo model -- keeps its own entries); rebuilt whenever the weights are resolved again.
    std::unordered_map<const void*, const float*> fp8_scale;
    bool adaln_combine = true;         // ltx2_dit_set_option("adaln_combine"): round 4, see forward()
    bool text_kv_ahead = true;         // ltx2_dit_set_option("text_kv_ahead"): round 5, AudioVideo V2.3 -- the video stream's sigma-modulated text K / V of layer l are projected on the SIDE stream at the top of the layer
    hipEvent_t text_kv_ev = nullptr;   //   (they depend on the prompt and sigma only); the main stream waits on this event in front of its text cross-attention
    int fold_norms = 1;                // ltx2_dit_set_option("fold_norms"): round 6, see block_attention(): 0 = every norm a pass of its own (round 5), 1 = the text cross-attention's
                                       //   plain RMS pre-norm rides on attn1.to_out's epilogue
    bool fp8_compute = false;          // ltx2_dit_set_option("fp8_compute"): fp8-resident weights x per-token fp8 activations on the fp8 MFMA
    bool prepared = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // AudioVideo: the audio modality's (small, latency-bound) kernels run on a side stream beside the video
    // GEMMs, whose 224-tile grids leave CUs idle; fork/join by events (captured as parallel graph branches).
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> sync_ev;
    size_t sync_next = 0;
    // live HIP-event profiling of one GEMM kernel instantiation (bench.py roofline)
    int prof_epi = -2;                 // -2 off, -1 every GEMM, >= 0 one epilogue
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    double prof_flops = 0;
};

namespace {

// Carve the workspace; base == nullptr only computes the size.
long carve(ltx2_dit* c, char* base, int N, int S, int Na, int Sa, int per_token) {
    const long L = c->cfg.num_layers;
    long off = 0;
    auto take = [&](long bytes) {
        char* p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    c->sigmas_dev = (float*)take(4L * 64);
    for (int k = 0; k < (c->av ? 2 : 1); ++k) {
        Mod& m = c->m[k];
        const long n = k ? Na : N, s = k ? Sa : S;
        const long D = m.D, npad = align_up(n, 64), spad = align_up(s, 64);
        const long Cctx = c->cfg.caption_channels > 0 ? c->cfg.caption_channels : D;
        const long T = per_token ? n : 1, rows = c->v2 ? 9 : 6;
        m.x = (float*)take(4L * n * D);
        m.lat = (bf16*)take(2L * n * m.Cin);
        m.h = (bf16*)take(2L * n * D);
        m.h2 = (bf16*)take(c->av ? 2L * n * D : 0);
        m.qkv = (bf16*)take(2L * n * 3 * D);      // also the cross-modal Q / K,V projected from this modality's tokens
        m.vt = (bf16*)take(2L * D * npad);
        m.att = (bf16*)take(2L * n * D);
        m.ff = (bf16*)take(2L * n * 4 * D);
        m.glog = (float*)take(c->gated ? 4L * n * m.H * GATE_LOGIT_PARTS : 0);       // (room for the K-slice partial sums of gate_logits_parts_launch)
        m.kmask = (unsigned long long*)take(8L * (spad / 64));
        m.qss = (float*)take(4L * n * (D / 64));
        m.knq = (float*)take(4L * L * D);
        const bool fold_bufs = k == 0 && !c->av;
        m.rss_ld = align_up(n, 256) + 256;
        m.rss = (float*)take(fold_bufs ? 4L * (D / 256 + 1) * m.rss_ld : 0);
        m.a8 = (unsigned char*)take((c->fp8_compute && k == 0) ? n * 4 * D : 0);
        m.a8s = (float*)take((c->fp8_compute && k == 0) ? 4L * n : 0);
        m.sin_f = (float*)take(4L * 256);
        m.e1_f = (float*)take(4L * D);
        m.aux_e = (float*)take(4L * D);
        m.e_f = (float*)take(4L * T * D);
        m.emb = (float*)take(4L * T * rows * D);
        m.prompt_emb = (float*)take(c->v2 ? 4L * 2 * D : 0);
        // round 4: every layer's scale_shift_table in one block (copied at prepare) and its per-step sum with the timestep embedding (adaln_combine)
        m.sst_all = (float*)take(4L * L * rows * D);
        m.comb = (float*)take(4L * L * rows * D);
        m.cross_ss = (float*)take(c->av ? 4L * 4 * D : 0);
        m.cross_gate = (float*)take(c->av ? 4L * D : 0);        // (directly behind cross_ss: the five cross-modal embedding rows are one block, see forward())
        m.ca_all = (float*)take(c->av ? 4L * L * 5 * D : 0);
        m.ca_comb = (float*)take(c->av ? 4L * L * 5 * D : 0);
        m.ts_tok = (float*)take(per_token ? 4L * n : 0);
        m.sin_b = (bf16*)take(per_token ? 2L * n * 256 : 0);
        m.e1_b = (bf16*)take(per_token ? 2L * n * D : 0);
        m.es_b = (bf16*)take(per_token ? 2L * n * D : 0);
        m.vel = (float*)take(4L * n * m.Cout);
        m.x0 = (float*)take(4L * n * m.Cout);
        m.cosb = (float*)take(4L * n * (D / 2));
        m.sinb = (float*)take(4L * n * (D / 2));
        const long Dc = c->m[c->av ? 1 : 0].D;         // cross-modal RoPE lives in the audio inner dim
        m.ccos = (float*)take(c->av ? 4L * n * (Dc / 2) : 0);
        m.csin = (float*)take(c->av ? 4L * n * (Dc / 2) : 0);
        m.ctx_in = (bf16*)take(2L * s * Cctx);
        m.c1 = (bf16*)take(c->cfg.caption_channels > 0 ? 2L * s * D : 0);
        m.ctxp = (bf16*)take(c->cfg.caption_channels > 0 ? 2L * s * D : 0);
        m.ctxm = (bf16*)take(c->v2 ? 2L * s * D : 0);
        const long kv_layers = c->v2 ? 1 : L;          // sigma-modulated text K/V cannot be cached per prompt
        m.kv2 = (bf16*)take(2L * kv_layers * s * 2 * D);
        m.vt2 = (bf16*)take(2L * kv_layers * D * spad);
    }
    return off;
}

const void* find(ltx2_dit* c, const std::string& name, int dtype, long numel) {
    auto it = c->weights.find(name);
    if (it == c->weights.end()) {
        ltx2_set_error("dit: missing weight '%s'", name.c_str());
        return nullptr;
    }
    if (dtype == LTX2_DTYPE_BF16 && it->second.dtype == LTX2_DTYPE_FP8_E4M3FN && it->second.n == numel) {
        auto sc = c->weights.find(name + "_scale");
        if (sc == c->weights.end() || sc->second.dtype != LTX2_DTYPE_F32) {
            ltx2_set_error("dit: fp8-resident weight '%s' has no fp32 '%s_scale' vector", name.c_str(), name.c_str());
            return nullptr;
        }
        c->fp8_scale[it->second.p] = (const float*)sc->second.p;
        return it->second.p;
    }
    if (it->second.dtype != dtype || it->second.n != numel) {
        ltx2_set_error("dit: weight '%s' has dtype %d / numel %ld, expected dtype %d / numel %ld", name.c_str(),
                       it->second.dtype, it->second.n, dtype, numel);
        return nullptr;
    }
    return it->second.p;
}

#define GETW(dst, name, rows, cols)                                                                          \
    do {                                                                                                     \
        dst = (const bf16*)find(c, std::string(name) + ".weight", LTX2_DTYPE_BF16, (long)(rows) * (cols));   \
        if (!dst) return LTX2_E_STATE;                                                                       \
    } while (0)
#define GETB(dst, name, n)                                                                    \
    do {                                                                                      \
        dst = (const float*)find(c, std::string(name) + ".bias", LTX2_DTYPE_F32, (long)(n));  \
        if (!dst) return LTX2_E_STATE;                                                        \
    } while (0)
#define GETF(dst, name, n)                                                          \
    do {                                                                            \
        dst = (const float*)find(c, std::string(name), LTX2_DTYPE_F32, (long)(n));  \
        if (!dst) return LTX2_E_STATE;                                              \
    } while (0)

int resolve_ada(ltx2_dit* c, AdaW& a, const std::string& name, long D, int rows) {
    GETW(a.t1_w, name + ".emb.timestep_embedder.linear_1", D, 256);
    GETB(a.t1_b, name + ".emb.timestep_embedder.linear_1", D);
    GETW(a.t2_w, name + ".emb.timestep_embedder.linear_2", D, D);
    GETB(a.t2_b, name + ".emb.timestep_embedder.linear_2", D);
    GETW(a.lin_w, name + ".linear", rows * D, D);
    GETB(a.lin_b, name + ".linear", rows * D);
    a.rows = rows;
    return LTX2_OK;
}

// query width Dq, key/value source width Dc, attention inner width Di, `heads` gate logits
int resolve_attn(ltx2_dit* c, AttnW& a, const std::string& name, long Dq, long Dc, long Di, int heads, bool self) {
    if (self) {
        GETW(a.qkv_w, name + ".to_qkv", 3 * Di, Dq);
        GETB(a.qkv_b, name + ".to_qkv", 3 * Di);
    } else {
        GETW(a.q_w, name + ".to_q", Di, Dq);
        GETB(a.q_b, name + ".to_q", Di);
        GETW(a.kv_w, name + ".to_kv", 2 * Di, Dc);
        GETB(a.kv_b, name + ".to_kv", 2 * Di);
    }
    GETW(a.o_w, name + ".to_out.0", Dq, Di);
    GETB(a.o_b, name + ".to_out.0", Dq);
    GETF(a.qn, name + ".q_norm.weight", Di);
    GETF(a.kn, name + ".k_norm.weight", Di);
    if (c->gated) {
        GETW(a.g_w, name + ".to_gate_logits", heads, Dq);
        GETB(a.g_b, name + ".to_gate_logits", heads);
    }
    return LTX2_OK;
}

int resolve(ltx2_dit* c) {
    if (c->resolved) return LTX2_OK;
    c->fp8_scale.clear();
    const int nm = c->av ? 2 : 1;
    const int rows = c->v2 ? 9 : 6;
    for (int k = 0; k < nm; ++k) {
        const std::string pre = k ? "audio_" : "";
        const long D = c->m[k].D;
        ModW& w = c->mw[k];
        GETW(w.patch_w, pre + "patchify_proj", D, c->m[k].Cin);
        GETB(w.patch_b, pre + "patchify_proj", D);
        TRY(resolve_ada(c, w.ada, pre + "adaln_single", D, rows));
        if (c->v2) TRY(resolve_ada(c, w.prompt, pre + "prompt_adaln_single", D, 2));
        if (c->cfg.caption_channels > 0) {
            GETW(w.cap1_w, pre + "caption_projection.linear_1", D, c->cfg.caption_channels);
            GETB(w.cap1_b, pre + "caption_projection.linear_1", D);
            GETW(w.cap2_w, pre + "caption_projection.linear_2", D, D);
            GETB(w.cap2_b, pre + "caption_projection.linear_2", D);
        }
        GETF(w.sst_out, pre + "scale_shift_table", 2 * D);
        GETW(w.proj_w, pre + "proj_out", c->m[k].Cout, D);
        GETB(w.proj_b, pre + "proj_out", c->m[k].Cout);
    }
    if (c->av) {
        TRY(resolve_ada(c, c->mw[0].cross_ss, "av_ca_video_scale_shift_adaln_single", c->m[0].D, 4));
        TRY(resolve_ada(c, c->mw[0].cross_gate, "av_ca_a2v_gate_adaln_single", c->m[0].D, 1));
        TRY(resolve_ada(c, c->mw[1].cross_ss, "av_ca_audio_scale_shift_adaln_single", c->m[1].D, 4));
        TRY(resolve_ada(c, c->mw[1].cross_gate, "av_ca_v2a_gate_adaln_single", c->m[1].D, 1));
    }
    c->layers.assign(c->cfg.num_layers, LayerW());
    for (int i = 0; i < c->cfg.num_layers; ++i) {
        LayerW& w = c->layers[i];
        const std::string p = "transformer_blocks." + std::to_string(i) + ".";
        for (int k = 0; k < nm; ++k) {
            const std::string pre = k ? "audio_" : "";
            const long D = c->m[k].D;
            BlockW& b = w.m[k];
            TRY(resolve_attn(c, b.self, p + pre + "attn1", D, D, D, c->m[k].H, true));
            TRY(resolve_attn(c, b.text, p + pre + "attn2", D, D, D, c->m[k].H, false));
            GETW(b.ff1_w, p + pre + "ff.net.0.proj", 4 * D, D);
            GETB(b.ff1_b, p + pre + "ff.net.0.proj", 4 * D);
            GETW(b.ff2_w, p + pre + "ff.net.2", D, 4 * D);
            GETB(b.ff2_b, p + pre + "ff.net.2", D);
            GETF(b.sst, p + pre + "scale_shift_table", rows * D);
            if (c->v2) GETF(b.prompt_sst, p + pre + "prompt_scale_shift_table", 2 * D);
        }
        if (c->av) {
            const long Dv = c->m[0].D, Da = c->m[1].D;
            TRY(resolve_attn(c, w.a2v, p + "audio_to_video_attn", Dv, Da, Da, c->m[1].H, false));
            TRY(resolve_attn(c, w.v2a, p + "video_to_audio_attn", Da, Dv, Da, c->m[1].H, false));
            GETF(w.ca[0], p + "scale_shift_table_a2v_ca_video", 5 * Dv);
            GETF(w.ca[1], p + "scale_shift_table_a2v_ca_audio", 5 * Da);
        }
    }
    c->resolved = true;
    return LTX2_OK;
}
#undef GETW
#undef GETB
#undef GETF

thread_local ltx2_dit* g_prof_ctx = nullptr;

// vt (optional): where the V third of a fused QKV projection goes as attention's V^T operand; *vt_done says whether the
// GEMM's epilogue wrote it (else the caller runs vt_transpose_launch on the V columns of `out`).
struct VtOut {
    bf16* vt;
    int col0, npad, hd;
};

// fp8 compute (opt-in): would dense() send this GEMM to the fp8 MFMA?  (the video stream's big GEMMs on fp8-resident weights)
bool f8_route(ltx2_dit* c, const bf16* W, int M, int N, int K, int epi) {
    if (!c->fp8_compute || !c->m[0].a8 || M < 1024 || M > c->m[0].N || K > 4 * c->m[0].D) return false;
    auto f8 = c->fp8_scale.find((const void*)W);
    if (f8 == c->fp8_scale.end()) return false;
    GemmParams q{};
    q.A8 = c->m[0].a8;
    q.ascale = c->m[0].a8s;
    q.lda = K;
    q.W8 = (const unsigned char*)W;
    q.wscale = f8->second;
    q.out = c->m[0].x;
    q.ldo = 8;
    q.M = M;
    q.N = N;
    q.K = K;
    return gemm_v4_f8_supported(q, epi);
}

// q_norm folded into the text cross-attention (round 3, VERDICT r2 #4c): possible when the query projection runs on a kernel whose
// epilogue writes the row partial sums of squares (the 4-wave kernel's bf16 epilogue: M >= 1024; D % 1024 == 0: the attention kernel's four lanes of a
// query row read the D / 64 partials as f32x4 quarters, so q_ss_ld = D / 64 must be a multiple of 16 -- ADVICE r5: D = 512 / 1536 / 2560 / 3584 fall back to the unfused q_norm pass)
bool text_qfold_ok(ltx2_dit* c, int k) {
    const Mod& m = c->m[k];
    if (m.D % 1024 != 0 || m.hd != 128) return false;
    const bf16* W = c->layers[0].m[k].text.q_w;
    GemmParams q{};
    q.M = m.N;
    q.N = m.D;
    q.K = m.D;
    q.lda = m.D;
    q.ldo = m.D;
    q.out = m.x;
    if (k == 0 && f8_route(c, W, m.N, m.D, m.D, EPI_BF16)) {
        q.A8 = m.a8;
        q.ascale = m.a8s;
    } else {
        q.A = m.h;
    }
    auto f8 = c->fp8_scale.find((const void*)W);
    if (f8 != c->fp8_scale.end()) {
        q.W8 = (const unsigned char*)W;
        q.wscale = f8->second;
    } else {
        q.W = W;
    }
    return gemm_rowss_supported(q, EPI_BF16);
}

// round 6: can the text cross-attention's pre-norm of this modality's blocks be folded around attn1.to_out / attn2.to_q (block_attention())?  VideoOnly non-V2.3
// ungated models on dense bf16 weights in the bfloat16 build (the un-normalised shadow is not safe in IEEE half), both GEMMs on the 4-wave layout-3 kernel
int fold_supported(ltx2_dit* c, int k) {
#ifdef LTX2_F16
    (void)c; (void)k;
    return 0;
#else
    if (c->av || c->v2 || c->gated || c->fp8_compute || !c->fp8_scale.empty() || k != 0) return 0;
    const Mod& m = c->m[k];
    if (!m.rss || m.D % 256 != 0 || m.D / 256 > GEMM_RF_MAX_PARTS) return 0;
    const BlockW& w = c->layers[0].m[k];
    GemmParams q{};
    q.A = m.h;
    q.lda = m.D;
    q.M = m.N;
    q.K = m.D;
    q.out = m.x;
    // producer: attn1.to_out; consumer: attn2.to_q
    q.W = w.self.o_w;
    q.N = m.D;
    q.ldo = m.D;
    q.shadow = m.h;
    q.ld_shadow = m.D;
    q.shadow_ss = m.rss;
    q.ld_ss = m.rss_ld;
    if (!gemm_fold_supported(q, EPI_RESID_GATE_F32)) return 0;
    GemmParams r{};
    r.A = m.h;
    r.lda = m.D;
    r.M = m.N;
    r.K = m.D;
    r.N = m.D;
    r.ldo = m.D;
    r.out = m.qkv;
    r.W = w.text.q_w;
    r.rf_parts = m.rss;
    r.rf_ld = m.rss_ld;
    r.rf_nparts = m.D / 256;
    r.rf_dim = m.D;
    if (!gemm_fold_supported(r, EPI_BF16)) return 0;
    return 1;
#endif
}

__global__ void vec_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// A folded norm's share of one GEMM (GemmParams, "round 6"): producer fields for a gated-residual GEMM, consumer fields for the projection behind it
struct Fold {
    bf16* shadow = nullptr;
    const float* shadow_scale = nullptr;
    float* shadow_ss = nullptr;
    long ld_shadow = 0;
    const float* rf_parts = nullptr;
    long ld_ss = 0;         // row stride of shadow_ss / rf_parts
    int rf_nparts = 0, rf_dim = 0;
    float rf_eps = 0.f;
};

// preq: the kernel that produced A already left its per-token e4m3fn codes + scales in m[0].a8 / a8s (norm_mod_launch's q8 output);
// only valid when f8_route() says yes for this GEMM
int dense(ltx2_dit* c, const bf16* A, long lda, const bf16* W, const float* bias, void* out, long ldo, int M, int N, int K, int epi,
          hipStream_t st, const float* gate = nullptr, long gate_stride = 0, const float* gate_table = nullptr,
          const VtOut* vt = nullptr, bool* vt_done = nullptr, bool preq = false, float* rowss = nullptr, const Fold* fold = nullptr) {
    GemmParams p{};
    if (fold) {
        p.shadow = fold->shadow;
        p.shadow_scale = fold->shadow_scale;
        p.shadow_ss = fold->shadow_ss;
        p.ld_shadow = fold->ld_shadow;
        p.ld_ss = fold->ld_ss;
        p.rf_parts = fold->rf_parts;
        p.rf_ld = fold->ld_ss;
        p.rf_nparts = fold->rf_nparts;
        p.rf_dim = fold->rf_dim;
        p.rf_eps = fold->rf_eps;
    }
    p.A = A;
    p.lda = lda;
    p.W = W;
    if (!c->fp8_scale.empty()) {
        auto f8 = c->fp8_scale.find((const void*)W);
        if (f8 != c->fp8_scale.end()) {      // fp8-resident: codes + per-row scale, expanded inside the GEMM
            p.W = nullptr;
            p.W8 = (const unsigned char*)W;
            p.wscale = f8->second;
        }
    }
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate = gate;
    p.gate_stride = gate_stride;
    p.gate_table = gate_table;
    p.rowss = rowss;
    // fp8 compute (opt-in): the video stream's big GEMMs on fp8-resident weights take per-token-quantised activations and the fp8 MFMA
    if (p.W8 && f8_route(c, W, M, N, K, epi)) {
        if (!preq) TRY(quant_rows_fp8_launch(A, lda, M, K, c->m[0].a8, K, c->m[0].a8s, st));
        p.A8 = c->m[0].a8;
        p.ascale = c->m[0].a8s;
        p.lda = K;
        p.A = nullptr;
    } else if (preq) {
        ltx2_set_error("dit: a pre-quantised activation reached a GEMM that does not take the fp8 path");
        return LTX2_E_STATE;
    }
    if (vt) {
        p.vt = vt->vt;
        p.vt_col0 = vt->col0;
        p.vt_npad = vt->npad;
        p.vt_hd = vt->hd;
        p.vt_head_stride = (long)vt->hd * vt->npad;
        *vt_done = gemm_vt_fused(p, epi);
        if (!*vt_done) p.vt = nullptr;
    }
    ltx2_dit* pc = g_prof_ctx;
    const bool prof = pc && (pc->prof_epi == -1 || pc->prof_epi == epi);
    if (prof) {
        while (pc->prof_ev.size() < pc->prof_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return LTX2_E_HIP;
            pc->prof_ev.push_back(e);
        }
        (void)hipEventRecord(pc->prof_ev[pc->prof_used], st);
    }
    const int rc = gemm_launch(p, epi, false, st);
    if (prof) {
        (void)hipEventRecord(pc->prof_ev[pc->prof_used + 1], st);
        pc->prof_used += 2;
        pc->prof_flops += 2.0 * M * N * K;
    }
    return rc;
}

// timesteps of a conditioned step (reference pipelines/common.py:193-232 timesteps_from_mask): ts[n] = mask[n] * sigma, sigma read on the device
__global__ void mask_sigma_kernel(const float* __restrict__ mask, const float* __restrict__ sigma, float* __restrict__ ts, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ts[i] = mask[i] * sigma[0];
}

__global__ void silu_cast_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = f2bf(silu_f(in[i]));
}

// AdaLayerNormSingle over T rows of timesteps (ts[i*t_stride] * mult): emb [T][rows*D] (fp32) and,
// when e_out != null, the embedded timestep e [T][D] (model.py:113-140).
int adaln_chain(ltx2_dit* c, Mod& m, const AdaW& a, const float* ts, long t_stride, int T, float mult, float* emb, float* e_out,
                hipStream_t st) {
    const int D = m.D;
    if (T == 1) {
        float* e = e_out ? e_out : m.aux_e;
        TRY(timestep_sinusoid_launch(ts, 0, 0.f, mult, 1, 256, m.sin_f, nullptr, st));
        TRY(gemv_launch(m.sin_f, 256, a.t1_w, a.t1_b, m.e1_f, D, 1, D, 256, 0, 1, st));
        TRY(gemv_launch(m.e1_f, D, a.t2_w, a.t2_b, e, D, 1, D, D, 0, 0, st));
        TRY(gemv_launch(e, D, a.lin_w, a.lin_b, emb, (long)a.rows * D, 1, a.rows * D, D, 1, 0, st));
    } else {
        TRY(timestep_sinusoid_launch(ts, t_stride, 0.f, mult, T, 256, nullptr, m.sin_b, st));
        TRY(dense(c, m.sin_b, 256, a.t1_w, a.t1_b, m.e1_b, D, T, D, 256, EPI_SILU_BF16, st));
        TRY(dense(c, m.e1_b, D, a.t2_w, a.t2_b, e_out, D, T, D, D, EPI_F32, st));
        hipLaunchKernelGGL(silu_cast_kernel, dim3(2048), dim3(256), 0, st, e_out, m.es_b, (long)T * D);
        LTX2_CHECK_LAUNCH("silu_cast_kernel");
        TRY(dense(c, m.es_b, D, a.lin_w, a.lin_b, emb, (long)a.rows * D, T, a.rows * D, D, EPI_F32, st));
    }
    return LTX2_OK;
}

// per-head gate logits as gate_logits() left them: one [rows][H] array, or GATE_LOGIT_PARTS partial sums + the bias (summed in the attention kernel's epilogue)
struct GateRef {
    const float* p = nullptr;
    int parts = 1;
    const float* bias = nullptr;
};
int attend(const bf16* q, long ldq, const bf16* k, long ldk, const bf16* vt, int npad, bf16* out, long ldo, int nq,
           int nkv, int H, int hd, hipStream_t st, const float* q_ss = nullptr, float q_eps = 0.f,
           const unsigned long long* kmask = nullptr, GateRef gate = GateRef{}) {
    AttnParams a{};
    a.kmask = kmask;
    a.gate = gate.p;        // per-head gate logits [nq][H] (V2.3): out *= 2 sigmoid(.) in the kernel's epilogue
    a.gate_ld = H;
    a.gate_parts = gate.parts;
    a.gate_bias = gate.bias;
    if (q_ss) {         // q_norm as a per-row softmax scale from the projection's partial sums (text cross-attention)
        a.q_ss = q_ss;
        a.q_ss_ld = H * hd / 64;
        a.q_norm_dim = H * hd;
        a.q_eps = q_eps;
    }
    a.Q = q;
    a.ldq = ldq;
    a.K = k;
    a.ldk = ldk;
    a.VT = vt;
    a.vt_head_stride = (long)hd * npad;
    a.O = out;
    a.ldo = ldo;
    a.Nq = nq;
    a.Nkv = nkv;
    a.Npad = npad;
    a.H = H;
    a.head_dim = hd;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
    return attn_launch(a, st);
}

// One event per fork/join edge of a forward (or of a whole captured loop): the pool grows on demand and
// is rewound at the start of every eager call / capture, so no event is re-recorded inside a capture.
hipEvent_t next_event(ltx2_dit* c) {
    if (c->sync_next == c->sync_ev.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c->sync_ev.push_back(e);
    }
    return c->sync_ev[c->sync_next++];
}

// `to` continues after everything enqueued on `from` so far
int stream_after(ltx2_dit* c, hipStream_t from, hipStream_t to) {
    if (from == to) return LTX2_OK;
    hipEvent_t e = next_event(c);
    if (!e || hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
        ltx2_set_error("dit: stream fork/join failed: %s", hipGetErrorString(hipGetLastError()));
        return LTX2_E_HIP;
    }
    return LTX2_OK;
}

// split form of stream_after: record now (the point on `from` others may wait for), make `to` wait later
hipEvent_t event_record(ltx2_dit* c, hipStream_t from) {
    hipEvent_t e = next_event(c);
    if (!e || hipEventRecord(e, from) != hipSuccess) {
        ltx2_set_error("dit: event record failed: %s", hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    return e;
}

int event_wait(hipEvent_t e, hipStream_t to) {
    if (!e || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
        ltx2_set_error("dit: stream wait failed: %s", hipGetErrorString(hipGetLastError()));
        return LTX2_E_HIP;
    }
    return LTX2_OK;
}

// Per-head gates (attention.py:241-249): att[:, h*hd:(h+1)*hd] *= 2*sigmoid(x @ Wg^T + bg)[:, h]
// many rows: K-slice partial sums, added (with the bias) by the attention kernel's epilogue; few rows (the audio stream): the one-launch form
inline bool gate_in_parts(int rows, int Dq) { return rows >= 1024 && Dq % (GATE_LOGIT_PARTS * 32) == 0 && Dq <= 8192; }
int gate_logits(ltx2_dit* c, Mod& m, const AttnW& w, const bf16* xin, int Dq, int rows, int H, hipStream_t st) {
    if (!c->gated) return LTX2_OK;
    m.glog_parts = gate_in_parts(rows, Dq) ? GATE_LOGIT_PARTS : 1;
    m.glog_bias = w.g_b;
    if (m.glog_parts > 1) return gate_logits_parts_launch(xin, Dq, w.g_w, m.glog, rows, Dq, H, st);
    return gate_logits_launch(xin, Dq, w.g_w, w.g_b, m.glog, H, rows, Dq, H, st);
}

// the gate logits gate_logits() left for the attention that follows (null for ungated models): the kernel's epilogue applies 2 sigmoid(.) per head
GateRef glog(ltx2_dit* c, const Mod& m) { return c->gated ? GateRef{m.glog, m.glog_parts, m.glog_bias} : GateRef{}; }

// K (k_norm applied, optional RoPE) and V^T of a text / cross-modal context
int project_kv(ltx2_dit* c, const bf16* ctx, int rows, int Dc, const AttnW& w, int Di, int H, int hd, float eps, const float* cosb,
               const float* sinb, bf16* kv, bf16* vt, int npad, hipStream_t st, const float* k_weight = nullptr) {
    const VtOut vo{vt, Di, npad, hd};
    bool vt_done = false;           // V^T straight from the K/V GEMM's epilogue where the 4-wave kernel takes the shape
    TRY(dense(c, ctx, Dc, w.kv_w, w.kv_b, kv, 2 * Di, rows, 2 * Di, Dc, EPI_BF16, st, nullptr, 0, nullptr, &vo, &vt_done));
    const int offs[1] = {0};
    const float* wts[1] = {k_weight ? k_weight : w.kn};
    TRY(qknorm_rope_launch(kv, 2 * Di, rows, Di, hd, 1, offs, wts, eps, cosb, sinb, st));
    if (vt_done) return LTX2_OK;
    return vt_transpose_launch(kv + Di, 2 * Di, vt, rows, npad, H, st, hd);
}

// V2.3: the prompt-AdaLN-modulated text context of layer l and its K / V^T (transformer.py:427-455) -- a function of the prompt and sigma only
int text_kv(ltx2_dit* c, int k, int l, hipStream_t st) {
    Mod& m = c->m[k];
    const BlockW& w = c->layers[l].m[k];
    const int D = m.D;
    TRY(ctx_mod_launch(m.ctx, m.ctxm, m.S, D, w.prompt_sst + D, w.prompt_sst, m.prompt_emb + D, m.prompt_emb, st));
    return project_kv(c, m.ctxm, m.S, D, w.text, D, m.H, m.hd, c->cfg.norm_eps, nullptr, nullptr, m.kv2, m.vt2, m.Spad, st, m.qfold ? m.knq + (long)l * D : nullptr);
}

// ---- round 6: the text cross-attention's plain RMS pre-norm folded around the two GEMMs beside it ----------------------------------------------------
// rms_norm(x) in front of attn2.to_q (W, b) equals r (x W^T) + b with the row factor r = rsqrt(mean x^2 + eps).  attn1.to_out's gated-residual epilogue, which
// forms x, also leaves y = bf16(x) in `h` with the partial sums of squares of x (GemmParams::shadow), and the query projection takes (y, the partial sums -> r
// inside the kernel): the norm pass between the two GEMMs (15 us, 85 MB) is gone.  Same mathematics, another rounding order: y is rounded before the row factor
// instead of after it (both relative 2^-9).  Row-invariant gates only (one timestep per modality).
// The same algebra covers the two AdaLN-modulated norms (self-attention, feed-forward): the scale rides on the shadow, and the shift's product t W^T can be formed
// one step ahead by the projection itself from an extra operand row.  That was built and measured in this round (git history: "fold: its own kernel
// instantiations"): every kernel-level figure favoured it (three norm passes, 52 us per layer, for ~20 us of epilogue work), the step did not -- +0.3 ... +0.5 ms:
// the GEMMs on both sides run 4-5 % slower inside the step than their stand-alone A/B says (profiles/r06_fold_norms.md).  Only this one stayed.
// the two halves on a Fold: the producer writes x's shadow into h and its partial sums into rss; the consumer reads them back
inline void fold_produce(Fold& f, Mod& m, const float* scale) {
    f.shadow = m.h;
    f.ld_shadow = m.D;
    f.shadow_ss = m.rss;
    f.ld_ss = m.rss_ld;
    f.shadow_scale = scale;
}
inline void fold_consume(Fold& f, const Mod& m, float eps) {
    f.rf_parts = m.rss;
    f.ld_ss = m.rss_ld;
    f.rf_nparts = m.D / 256;
    f.rf_dim = m.D;
    f.rf_eps = eps;
}

// Self-attention, text cross-attention of one modality (transformer.py:503-554 / 191-226)
int block_attention(ltx2_dit* c, int k, int l, long es, hipStream_t st, int fold = 0) {
    Mod& m = c->m[k];
    const BlockW& w = c->layers[l].m[k];
    const int N = m.N, D = m.D, H = m.H, hd = m.hd;
    const float eps = c->cfg.norm_eps;
    // AdaLN rows: table row r + embedding row r; with one timestep per modality the sum was formed for every layer at the top of forward()
    const float* const tab = m.use_comb ? m.comb + (long)l * (c->v2 ? 9 : 6) * D : w.sst;
    const float* const emb0 = m.use_comb ? nullptr : m.emb;
    auto E = [&](int r) { return emb0 ? emb0 + (long)r * D : nullptr; };
    // self-attention: AdaLN rows (shift, scale, gate) = sst[0:3] + emb[0:3]
    // fp8 compute: the norm kernels hand the following GEMM per-token e4m3fn codes directly (and skip the bf16 copy nobody else reads)
    const bool q1 = k == 0 && f8_route(c, w.self.qkv_w, N, 3 * D, D, EPI_BF16);
    const int fl = k == 0 ? fold : 0;
    TRY(norm_mod_launch(m.x, D, (q1 && !c->gated) ? nullptr : m.h, D, N, D, eps, 0, tab + D, tab, E(1), E(0), es, st, q1 ? m.a8 : nullptr, D,
                        q1 ? m.a8s : nullptr));
    TRY(gate_logits(c, m, w.self, m.h, D, N, H, st));
    const VtOut vo{m.vt, 2 * D, m.Npad, hd};
    bool vt_done = false;           // the QKV GEMM's epilogue writes V^T itself where it can (gemm_v4.hip)
    TRY(dense(c, m.h, D, w.self.qkv_w, w.self.qkv_b, m.qkv, 3 * D, N, 3 * D, D, EPI_BF16, st, nullptr, 0, nullptr, &vo, &vt_done, q1));
    {
        const int offs[2] = {0, D};
        const float* wts[2] = {w.self.qn, w.self.kn};
        TRY(qknorm_rope_launch(m.qkv, 3 * D, N, D, hd, 2, offs, wts, eps, m.cosb, m.sinb, st));
    }
    if (!vt_done) TRY(vt_transpose_launch(m.qkv + 2 * D, 3 * D, m.vt, N, m.Npad, H, st, hd));
    TRY(attend(m.qkv, 3 * D, m.qkv + D, 3 * D, m.vt, m.Npad, m.att, D, N, N, H, hd, st, nullptr, 0.f, nullptr, glog(c, m)));
    Fold fo{};
    if (fl >= 1) fold_produce(fo, m, nullptr);          // the new residual rows also leave as the cross-attention query projection's operand (plain RMS norm: no scale, no shift)
    TRY(dense(c, m.att, D, w.self.o_w, w.self.o_b, m.x, D, N, D, D, EPI_RESID_GATE_F32, st, E(2), es, tab + 2 * D, nullptr, nullptr, false, nullptr, fl >= 1 ? &fo : nullptr));

    // text cross-attention: no RoPE, no mask.  V1: plain RMSNorm on x, K/V cached per prompt.
    // V2.3 (transformer.py:427-455): q side modulated by rows (6,7) and gated by row 8; the context
    // is modulated by the prompt AdaLN (rows shift, scale), so K/V are recomputed every step.
    const bf16* kk;
    const bf16* vt;
    const bool q2 = k == 0 && f8_route(c, w.text.q_w, N, D, D, EPI_BF16);
    bf16* h2 = (q2 && !c->gated) ? nullptr : m.h;
    unsigned char* a8q = q2 ? m.a8 : nullptr;
    float* a8sq = q2 ? m.a8s : nullptr;
    if (c->v2) {
        TRY(norm_mod_launch(m.x, D, h2, D, N, D, eps, 0, tab + 7 * D, tab + 6 * D, E(7), E(6), es, st, a8q, D, a8sq));
        if (k == 0 && c->text_kv_ev) {      // projected ahead on the side stream (forward()): wait for it here
            TRY(event_wait(c->text_kv_ev, st));
            c->text_kv_ev = nullptr;
        } else {
            TRY(text_kv(c, k, l, st));
        }
        kk = m.kv2;
        vt = m.vt2;
    } else {
        if (fl < 1) TRY(norm_mod_launch(m.x, D, h2, D, N, D, eps, 0, nullptr, nullptr, nullptr, nullptr, 0, st, a8q, D, a8sq));
        kk = m.kv2 + (long)l * m.S * 2 * D;
        vt = m.vt2 + (long)l * D * m.Spad;
    }
    TRY(gate_logits(c, m, w.text, m.h, D, N, H, st));
    // q_norm folded (m.qfold): the projection's epilogue leaves the partial sums of squares of its rows; q_norm.weight already sits on the keys
    // (k_norm.weight * q_norm.weight, prepare) and the row's RMS factor becomes its softmax scale -- no pass over q between GEMM and attention
    Fold fc{};
    fold_consume(fc, m, eps);
    TRY(dense(c, m.h, D, w.text.q_w, w.text.q_b, m.qkv, D, N, D, D, EPI_BF16, st, nullptr, 0, nullptr, nullptr, nullptr, q2, m.qfold ? m.qss : nullptr, fl >= 1 ? &fc : nullptr));
    if (!m.qfold) {
        const int offs[1] = {0};
        const float* wts[1] = {w.text.qn};
        TRY(qknorm_rope_launch(m.qkv, D, N, D, hd, 1, offs, wts, eps, nullptr, nullptr, st));
    }
    TRY(attend(m.qkv, D, kk, 2 * D, vt, m.Spad, m.att, D, N, m.S, H, hd, st, m.qfold ? m.qss : nullptr, eps,
               m.has_kmask ? m.kmask : nullptr, glog(c, m)));
    if (c->v2)
        TRY(dense(c, m.att, D, w.text.o_w, w.text.o_b, m.x, D, N, D, D, EPI_RESID_GATE_F32, st, E(8), es, tab + 8 * D));
    else
        TRY(dense(c, m.att, D, w.text.o_w, w.text.o_b, m.x, D, N, D, D, EPI_RESID_GATE_F32, st));
    return LTX2_OK;
}

// Feed-forward: AdaLN rows 3..5 (transformer.py:229-236,622-642); Linear -> GELU(tanh) -> Linear
int block_ffn(ltx2_dit* c, int k, int l, long es, hipStream_t st) {
    Mod& m = c->m[k];
    const BlockW& w = c->layers[l].m[k];
    const int N = m.N, D = m.D;
    const float* const tab = m.use_comb ? m.comb + (long)l * (c->v2 ? 9 : 6) * D : w.sst;
    const float* const emb0 = m.use_comb ? nullptr : m.emb;
    auto E = [&](int r) { return emb0 ? emb0 + (long)r * D : nullptr; };
    const bool q3 = k == 0 && f8_route(c, w.ff1_w, N, 4 * D, D, EPI_GELU_BF16);
    TRY(norm_mod_launch(m.x, D, q3 ? nullptr : m.h, D, N, D, c->cfg.norm_eps, 0, tab + 4 * D, tab + 3 * D, E(4), E(3), es, st,
                        q3 ? m.a8 : nullptr, D, q3 ? m.a8s : nullptr));
    TRY(dense(c, m.h, D, w.ff1_w, w.ff1_b, m.ff, 4 * D, N, 4 * D, D, EPI_GELU_BF16, st, nullptr, 0, nullptr, nullptr, nullptr, q3));
    TRY(dense(c, m.ff, 4 * D, w.ff2_w, w.ff2_b, m.x, D, N, D, 4 * D, EPI_RESID_GATE_F32, st, E(5), es, tab + 5 * D));
    return LTX2_OK;
}

// Audio <-> video cross-modal attention (transformer.py:556-620).  Table rows
// (scale_a2v, shift_a2v, scale_v2a, shift_v2a, gate); both directions read the SAME pre-update
// RMS-normalised streams, so all four modulated inputs are formed before x is touched.
//
// Round 4 schedule (was: 23 kernels in one chain on the main stream, ~500 us per layer with the chip mostly idle).  Only the audio -> video
// direction feeds the video stream (its feed-forward reads the a2v update); the video -> audio direction updates the AUDIO stream only.  So:
//   main (video): ONE norm launch for both modulations of vx -> a2v query projection -> rope(q) -> [audio K/V ready] -> a2v attention ->
//                 out-projection into vx -> (caller) video feed-forward
//   side (audio): ONE norm launch for both modulations of ax -> a2v K/V from audio (skinny) -> [signal] -> v2a query (skinny) -> [vx modulations
//                 ready] -> v2a key / value projection (the other big video-side GEMM) -> k-norm + rope -> v2a attention -> out-projection into ax ->
//                 (caller) audio feed-forward
// Buffers: the video modality's qkv scratch holds the a2v queries [N][Da] and, behind them, the v2a keys / values [N][2 Da]; the audio modality's
// holds the a2v keys / values [Na][2 Da] and, behind them, the v2a queries [Na][Da] -- the two directions never share a region.
int block_cross_modal(ltx2_dit* c, int l, hipStream_t st, hipStream_t sa) {
    Mod &v = c->m[0], &a = c->m[1];
    const LayerW& w = c->layers[l];
    const int Dv = v.D, Da = a.D, H = a.H, hd = a.hd;
    const float eps = c->cfg.norm_eps;
    const bool comb = c->adaln_combine;          // table + embedding rows already summed for every layer (forward())
    const float *tv = comb ? v.ca_comb + (long)l * 5 * Dv : w.ca[0], *ta = comb ? a.ca_comb + (long)l * 5 * Da : w.ca[1];
    const int offs[1] = {0};
    bf16* v_q = v.qkv;                              // a2v queries  [N][Da]
    bf16* v_kv = v.qkv + (long)v.N * Da;            // v2a keys | values [N][2 Da]
    bf16* a_kv = a.qkv;                             // a2v keys | values [Na][2 Da]
    bf16* a_q = a.qkv + (long)a.N * 2 * Da;         // v2a queries [Na][Da]
    // Capture order matters under hipGraph: a node's FIRST captured child continues its stream's run list (ROCm executes a run list serially), so
    // after every event record the recording stream's own next kernel is enqueued before the other stream's wait + consumer.
    // ---- video side, main stream: both modulations of vx, the two video-side projections ----
    {
        const float* sct[2] = {tv, tv + 2 * Dv};
        const float* sht[2] = {tv + Dv, tv + 3 * Dv};
        const float* sce[2] = {comb ? nullptr : v.cross_ss, comb ? nullptr : v.cross_ss + 2 * Dv};
        const float* she[2] = {comb ? nullptr : v.cross_ss + Dv, comb ? nullptr : v.cross_ss + 3 * Dv};
        TRY(norm_mod2_launch(v.x, Dv, v.h, v.h2, Dv, v.N, Dv, eps, sct, sht, sce, she, st));       // a2v query side | v2a context side
    }
    // the v2a key / value projection (the other big video-side GEMM) runs on the side stream from here, beside the main stream's a2v chain: same-box
    // A/B of the LTX-2.3 step, hipGraph: 104.0 ms against 106.0 with both projections in ONE launch on the main stream (128 + 256 tiles are two rounds
    // of the 256 CUs whichever way they are launched; on two streams the second round overlaps the a2v chain's small kernels)
    bool vt_done = false;
    const VtOut vo{v.vt, Da, v.Npad, hd};
    hipEvent_t video_norm = event_record(c, st);
    if (!video_norm) return LTX2_E_HIP;
    TRY(gate_logits(c, v, w.a2v, v.h, Dv, v.N, H, st));
    TRY(dense(c, v.h, Dv, w.a2v.q_w, w.a2v.q_b, v_q, Da, v.N, Da, Dv, EPI_BF16, st));
    {
        const float* wts[1] = {w.a2v.qn};
        TRY(qknorm_rope_launch(v_q, Da, v.N, Da, hd, 1, offs, wts, eps, v.ccos, v.csin, st));
    }
    // ---- audio side, side stream: both modulations of ax, a2v K / V, v2a Q ----
    {
        const float* sct[2] = {ta, ta + 2 * Da};
        const float* sht[2] = {ta + Da, ta + 3 * Da};
        const float* sce[2] = {comb ? nullptr : a.cross_ss, comb ? nullptr : a.cross_ss + 2 * Da};
        const float* she[2] = {comb ? nullptr : a.cross_ss + Da, comb ? nullptr : a.cross_ss + 3 * Da};
        TRY(norm_mod2_launch(a.x, Da, a.h, a.h2, Da, a.N, Da, eps, sct, sht, sce, she, sa));       // a2v context side | v2a query side
    }
    TRY(project_kv(c, a.h, a.N, Da, w.a2v, Da, H, hd, eps, a.ccos, a.csin, a_kv, a.vt, a.Npad, sa));
    hipEvent_t audio_kv = event_record(c, sa);
    if (!audio_kv) return LTX2_E_HIP;
    TRY(gate_logits(c, a, w.v2a, a.h2, Da, a.N, H, sa));
    TRY(dense(c, a.h2, Da, w.v2a.q_w, w.v2a.q_b, a_q, Da, a.N, Da, Da, EPI_BF16, sa));
    {
        const float* wts[1] = {w.v2a.qn};
        TRY(qknorm_rope_launch(a_q, Da, a.N, Da, hd, 1, offs, wts, eps, a.ccos, a.csin, sa));
    }
    // ---- audio -> video attention (main): Q from video, K / V from audio ----
    TRY(event_wait(audio_kv, st));
    TRY(attend(v_q, Da, a_kv, 2 * Da, a.vt, a.Npad, v.att, Da, v.N, a.N, H, hd, st, nullptr, 0.f, nullptr, glog(c, v)));
    TRY(dense(c, v.att, Da, w.a2v.o_w, w.a2v.o_b, v.x, Dv, v.N, Dv, Da, EPI_RESID_GATE_F32, st, comb ? nullptr : v.cross_gate, 0, tv + 4 * Dv));
    // ---- video -> audio attention (side): Q from audio, K / V from video ----
    TRY(event_wait(video_norm, sa));
    TRY(dense(c, v.h2, Dv, w.v2a.kv_w, w.v2a.kv_b, v_kv, 2 * Da, v.N, 2 * Da, Dv, EPI_BF16, sa, nullptr, 0, nullptr, &vo, &vt_done));
    {
        const float* wts[1] = {w.v2a.kn};
        TRY(qknorm_rope_launch(v_kv, 2 * Da, v.N, Da, hd, 1, offs, wts, eps, v.ccos, v.csin, sa));
    }
    if (!vt_done) TRY(vt_transpose_launch(v_kv + Da, 2 * Da, v.vt, v.N, v.Npad, H, sa, hd));
    TRY(attend(a_q, Da, v_kv, 2 * Da, v.vt, v.Npad, a.att, Da, a.N, v.N, H, hd, sa, nullptr, 0.f, nullptr, glog(c, a)));     // few queries, long KV (side stream)
    TRY(dense(c, a.att, Da, w.v2a.o_w, w.v2a.o_b, a.x, Da, a.N, Da, Da, EPI_RESID_GATE_F32, sa, comb ? nullptr : a.cross_gate, 0, ta + 4 * Da));
    return LTX2_OK;
}

struct ModIn {
    const float* latent;
    const float* ts;       // n_ts timesteps (device)
    int n_ts;
    const float* sigma;    // 1 float (device): this modality's scalar sigma (prompt / cross AdaLN)
    float* velocity;
};

int forward(ltx2_dit* c, const ModIn* in, hipStream_t st, bool rewind_events = true) {
    const int nm = c->av ? 2 : 1;
    if (rewind_events) c->sync_next = 0;
    long es[2] = {0, 0}, ee[2] = {0, 0};
    for (int k = 0; k < nm; ++k) {
        Mod& m = c->m[k];
        const ModW& w = c->mw[k];
        LTX2_CHECK_ARG(in[k].n_ts == 1 || in[k].n_ts == m.N, "dit_forward: n_timesteps=%d must be 1 or N=%d", in[k].n_ts, m.N);
        LTX2_CHECK_ARG(in[k].n_ts == 1 || c->per_token, "dit_forward: workspace was not sized for per-token timesteps");
        // patchify_proj (model.py:242) -> fp32 residual stream
        TRY(cast_f32_bf16_launch(in[k].latent, m.lat, (long)m.N * m.Cin, st));
        TRY(dense(c, m.lat, m.Cin, w.patch_w, w.patch_b, m.x, m.D, m.N, m.D, m.Cin, EPI_F32, st));
        // AdaLN-single (model.py:113-140)
        TRY(adaln_chain(c, m, w.ada, in[k].ts, 1, in[k].n_ts, c->cfg.timestep_scale, m.emb, m.e_f, st));
        if (in[k].n_ts > 1) {
            es[k] = (long)w.ada.rows * m.D;
            ee[k] = m.D;
        } else if (c->adaln_combine) {
            // one timestep for the whole modality (every loop but the conditioned ones): table + embedding of all layers in one launch; the
            // blocks then pass the sums as tables (rows_of()), the norm kernels read half the vectors and the gated-residual GEMMs no per-row gate
            TRY(adaln_combine_launch(m.sst_all, m.emb, m.comb, c->cfg.num_layers, (long)w.ada.rows * m.D, st));
        }
        m.use_comb = in[k].n_ts == 1 && c->adaln_combine;
        if (c->v2)           // prompt AdaLN from this modality's sigma (model.py:151-161)
            TRY(adaln_chain(c, m, w.prompt, in[k].sigma, 0, 1, c->cfg.timestep_scale, m.prompt_emb, nullptr, st));
        if (c->av) {         // cross-modal AdaLN from the OTHER modality's sigma (model.py:346-364,392-404)
            const float* cs = in[1 - k].sigma;
            TRY(adaln_chain(c, m, w.cross_ss, cs, 0, 1, c->cfg.timestep_scale, m.cross_ss, nullptr, st));
            TRY(adaln_chain(c, m, w.cross_gate, cs, 0, 1, c->cfg.av_ca_timestep_scale, m.cross_gate, nullptr, st));
            if (c->adaln_combine) {      // the five cross-modal rows (scale / shift x 2, gate) of every layer + their embeddings (cross_ss | cross_gate are adjacent)
                LTX2_CHECK_ARG(m.cross_gate == m.cross_ss + 4L * m.D, "dit_forward: cross-modal embedding rows are not contiguous");
                TRY(adaln_combine_launch(m.ca_all, m.cross_ss, m.ca_comb, c->cfg.num_layers, 5L * m.D, st));
            }
        }
    }
    // round 6: the text cross-attention's pre-norm folded around its GEMMs (block_attention()): row-invariant gates only
    const int fold = (!c->av && c->m[0].use_comb && c->fold_norms >= 1 && c->m[0].fold_max >= 1) ? 1 : 0;
    // the audio modality's block program runs on the side stream, joined around the cross-modal attention
    hipStream_t sa = (c->av && c->side) ? c->side : st;
    c->text_kv_ev = nullptr;
    for (int l = 0; l < c->cfg.num_layers; ++l) {
        if (c->av) {
            TRY(stream_after(c, st, sa));
            // the side stream has just waited for everything of layer l - 1 (its text cross-attention read kv2 / vt2: free now); the video stream's
            // text K / V of THIS layer go first on it, beside the main stream's norm / gate / QKV projection
            // (not with fp8 compute: that projection quantises its input into the video modality's ONE activation scratch, which the main stream is using)
            if (c->v2 && c->text_kv_ahead && sa != st && !c->fp8_compute) {
                TRY(text_kv(c, 0, l, sa));
                c->text_kv_ev = event_record(c, sa);
                if (!c->text_kv_ev) return LTX2_E_HIP;
            }
            TRY(block_attention(c, 1, l, es[1], sa));
        }
        TRY(block_attention(c, 0, l, es[0], st, fold));
        if (c->av) {
            TRY(block_cross_modal(c, l, st, sa));      // main: the video side; side: the audio side (events inside)
            TRY(block_ffn(c, 1, l, es[1], sa));
        }
        TRY(block_ffn(c, 0, l, es[0], st));
        if (c->av) TRY(stream_after(c, sa, st));
    }
    // output heads (model.py:744-774): LayerNorm(no affine) * (1 + scale) + shift, rows (shift, scale)
    for (int k = 0; k < nm; ++k) {
        Mod& m = c->m[k];
        const ModW& w = c->mw[k];
        TRY(norm_mod_launch(m.x, m.D, m.h, m.D, m.N, m.D, c->cfg.norm_eps, 1, w.sst_out + m.D, w.sst_out, m.e_f, m.e_f, ee[k], st));
        TRY(dense(c, m.h, m.D, w.proj_w, w.proj_b, in[k].velocity, m.Cout, m.N, m.Cout, m.D, EPI_F32, st));
    }
    return LTX2_OK;
}

struct StepIo {
    float* latent;
    const float* mask;
    const float* clean;
    float* x0_out;
};

int denoise_step(ltx2_dit* c, ModIn* in, const StepIo* io, float sigma, float sigma_next, hipStream_t st,
                 bool rewind_events = true) {
    const int nm = c->av ? 2 : 1;
    for (int k = 0; k < nm; ++k) {
        in[k].latent = io[k].latent;
        in[k].velocity = c->m[k].vel;
    }
    TRY(forward(c, in, st, rewind_events));
    for (int k = 0; k < nm; ++k) {
        Mod& m = c->m[k];
        float* x0 = io[k].x0_out ? io[k].x0_out : m.x0;
        TRY(x0_from_velocity_launch(io[k].latent, m.vel, in[k].ts, in[k].n_ts == 1 ? 0 : 1, 0.f, x0, m.N, m.Cout, st));
        TRY(euler_step_launch(io[k].latent, x0, io[k].mask, io[k].clean, sigma, sigma_next, io[k].latent, m.N, m.Cout, st));
    }
    return LTX2_OK;
}

int prepare_modality(ltx2_dit* c, int k, const float* context, int S, const float* cosb, const float* sinb,
                     const float* ccos, const float* csin, hipStream_t st) {
    Mod& m = c->m[k];
    const ModW& w = c->mw[k];
    LTX2_CHECK_ARG(context && cosb && sinb, "dit_prepare: null argument");
    LTX2_CHECK_ARG(S == m.S, "dit_prepare: S=%d differs from the bound workspace S=%d", S, m.S);
    const int D = m.D;
    const int Cctx = c->cfg.caption_channels > 0 ? c->cfg.caption_channels : D;
    bool ok = hipMemcpyAsync(m.cosb, cosb, 4L * m.N * (D / 2), hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipMemcpyAsync(m.sinb, sinb, 4L * m.N * (D / 2), hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (c->av) {
        LTX2_CHECK_ARG(ccos && csin, "dit_prepare: cross-modal RoPE tables are required for AudioVideo models");
        const long Dc = c->m[1].D;
        ok = ok && hipMemcpyAsync(m.ccos, ccos, 4L * m.N * (Dc / 2), hipMemcpyDeviceToDevice, st) == hipSuccess &&
             hipMemcpyAsync(m.csin, csin, 4L * m.N * (Dc / 2), hipMemcpyDeviceToDevice, st) == hipSuccess;
    }
    if (!ok) {
        ltx2_set_error("dit_prepare: RoPE table copy failed");
        return LTX2_E_HIP;
    }
    TRY(cast_f32_bf16_launch(context, m.ctx_in, (long)S * Cctx, st));
    m.ctx = m.ctx_in;
    if (c->cfg.caption_channels > 0) {   // PixArtAlphaTextProjection (model.py:52-56)
        TRY(dense(c, m.ctx_in, Cctx, w.cap1_w, w.cap1_b, m.c1, D, S, D, Cctx, EPI_GELU_BF16, st));
        TRY(dense(c, m.c1, D, w.cap2_w, w.cap2_b, m.ctxp, D, S, D, D, EPI_BF16, st));
        m.ctx = m.ctxp;
    }
    {   // every layer's scale_shift_table side by side: forward() adds the step's timestep embedding to all of them in one launch
        const long n = (long)(c->v2 ? 9 : 6) * D;
        for (int l = 0; l < c->cfg.num_layers; ++l)
            if (hipMemcpyAsync(m.sst_all + l * n, c->layers[l].m[k].sst, 4L * n, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                ltx2_set_error("dit_prepare: scale_shift_table copy failed");
                return LTX2_E_HIP;
            }
    }
    if (c->av)
        for (int l = 0; l < c->cfg.num_layers; ++l)
            if (hipMemcpyAsync(m.ca_all + (long)l * 5 * D, c->layers[l].ca[k], 4L * 5 * D, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                ltx2_set_error("dit_prepare: cross-modal scale_shift_table copy failed");
                return LTX2_E_HIP;
            }
    m.fold_max = fold_supported(c, k);
    m.qfold = text_qfold_ok(c, k);
    if (m.qfold)
        for (int l = 0; l < c->cfg.num_layers; ++l) {
            const AttnW& tw = c->layers[l].m[k].text;
            hipLaunchKernelGGL(vec_mul_kernel, dim3((D + 255) / 256), dim3(256), 0, st, tw.kn, tw.qn, m.knq + (long)l * D, D);
        }
    if (!c->v2)
        for (int l = 0; l < c->cfg.num_layers; ++l)
            TRY(project_kv(c, m.ctx, S, D, c->layers[l].m[k].text, D, m.H, m.hd, c->cfg.norm_eps, nullptr, nullptr,
                           m.kv2 + (long)l * S * 2 * D, m.vt2 + (long)l * D * m.Spad, m.Spad, st, m.qfold ? m.knq + (long)l * D : nullptr));
    return LTX2_OK;
}

int begin_capture(ltx2_dit* c, const float* host_sigmas, int n_steps, hipStream_t st) {
    LTX2_CHECK_ARG(st != nullptr, "dit_graph_capture: needs a non-default stream");
    for (int i = 0; i < n_steps; ++i) LTX2_CHECK_ARG(host_sigmas[i] != 0.f, "Sigma can't be 0.0");
    if (hipMemcpyAsync(c->sigmas_dev, host_sigmas, 4L * (n_steps + 1), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        ltx2_set_error("dit_graph_capture: sigma upload failed");
        return LTX2_E_HIP;
    }
    if (c->exec) {
        (void)hipGraphExecDestroy(c->exec);
        c->exec = nullptr;
    }
    if (c->graph) {
        (void)hipGraphDestroy(c->graph);
        c->graph = nullptr;
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        ltx2_set_error("dit_graph_capture: hipStreamBeginCapture failed");
        return LTX2_E_HIP;
    }
    return LTX2_OK;
}

int end_capture(ltx2_dit* c, int rc, hipStream_t st) {
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != LTX2_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        ltx2_set_error("dit_graph_capture: hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return LTX2_E_HIP;
    }
    c->graph = g;
    if (hipGraphInstantiate(&c->exec, g, nullptr, nullptr, 0) != hipSuccess) {
        ltx2_set_error("dit_graph_capture: hipGraphInstantiate failed");
        return LTX2_E_HIP;
    }
    return LTX2_OK;
}

int check_ready(ltx2_dit* c, const char* who, bool want_av) {
    LTX2_CHECK_ARG(c, "%s: null context", who);
    LTX2_CHECK_ARG(c->av == want_av, "%s: model is %s", who, c->av ? "AudioVideo (use the *_av entry points)" : "VideoOnly");
    if (!c->prepared) {
        ltx2_set_error("%s: ltx2_dit_prepare has not been called", who);
        return LTX2_E_STATE;
    }
    return LTX2_OK;
}

int bind(ltx2_dit* c, void* ptr, int64_t bytes, int N, int S, int Na, int Sa, int per_token) {
    LTX2_CHECK_ARG(((uintptr_t)ptr & 255) == 0, "dit_bind_workspace: pointer must be 256-byte aligned");
    const long need = carve(c, (char*)ptr, N, S, Na, Sa, per_token);
    if (bytes < need) {
        ltx2_set_error("dit_bind_workspace: %ld bytes given, %ld needed", (long)bytes, need);
        return LTX2_E_STATE;
    }
    c->ws = (char*)ptr;
    c->ws_bytes = bytes;
    const int n[2] = {N, Na}, s[2] = {S, Sa};
    for (int k = 0; k < 2; ++k) {
        c->m[k].N = n[k];
        c->m[k].S = s[k];
        c->m[k].Npad = (int)align_up(n[k], 64);
        c->m[k].Spad = (int)align_up(s[k], 64);
    }
    c->per_token = per_token;
    c->prepared = false;
    c->m[0].has_kmask = c->m[1].has_kmask = false;
    return LTX2_OK;
}

}  // namespace

extern "C" {

int ltx2_dit_create(const ltx2_dit_config* cfg, ltx2_dit** out) {
    LTX2_CHECK_ARG(cfg && out, "dit_create: null argument");
    LTX2_CHECK_ARG(cfg->head_dim == 128 || cfg->head_dim == 64, "dit_create: head_dim=%d, only 128 and 64 are implemented", cfg->head_dim);
    LTX2_CHECK_ARG(cfg->num_layers > 0 && cfg->num_heads > 0, "dit_create: bad layer/head count");
    LTX2_CHECK_ARG(cfg->in_channels % 64 == 0, "dit_create: in_channels must be a multiple of 64");
    LTX2_CHECK_ARG(cfg->caption_channels % 64 == 0, "dit_create: caption_channels must be a multiple of 64");
    LTX2_CHECK_ARG(cfg->model_type == LTX2_MODEL_VIDEO_ONLY || cfg->model_type == LTX2_MODEL_AUDIO_VIDEO, "dit_create: bad model_type %d", cfg->model_type);
    ltx2_dit* c = new ltx2_dit();
    c->cfg = *cfg;
    c->av = cfg->model_type == LTX2_MODEL_AUDIO_VIDEO;
    c->v2 = cfg->cross_attention_adaln != 0;
    c->gated = cfg->apply_gated_attention != 0;
    Mod& v = c->m[0];
    v.H = cfg->num_heads;
    v.hd = cfg->head_dim;
    v.D = v.H * v.hd;
    v.Cin = cfg->in_channels;
    v.Cout = cfg->out_channels;
    if (c->av) {
        if (!(cfg->audio_head_dim == 64 && cfg->audio_heads == cfg->num_heads && cfg->audio_in_channels % 64 == 0)) {
            delete c;
            ltx2_set_error("dit_create: AudioVideo needs audio_head_dim = 64, audio_heads = num_heads (shared cross-modal RoPE heads) and audio_in_channels %% 64 == 0");
            return LTX2_E_INVALID;
        }
        Mod& a = c->m[1];
        a.H = cfg->audio_heads;
        a.hd = cfg->audio_head_dim;
        a.D = a.H * a.hd;
        a.Cin = cfg->audio_in_channels;
        a.Cout = cfg->audio_out_channels;
    }
    if (c->av) {
        // (round 4: a high-priority side stream measured the same as the default priority, eager and under hipGraph; a LOW priority one 144 vs 105 ms
        // eager -- the audio chain then starves behind the video GEMMs' queued tiles)
        const bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
        if (!ok) {
            ltx2_set_error("dit_create: side stream / event creation failed");
            return LTX2_E_HIP;
        }
    }
    *out = c;
    return LTX2_OK;
}

void ltx2_dit_destroy(ltx2_dit* c) {
    if (!c) return;
    for (hipEvent_t e : c->sync_ev) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->exec) (void)hipGraphExecDestroy(c->exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    delete c;
}

int ltx2_dit_set_weight(ltx2_dit* c, const char* name, const void* ptr, int dtype, int64_t numel) {
    LTX2_CHECK_ARG(c && name && ptr, "dit_set_weight: null argument");
    LTX2_CHECK_ARG(dtype == LTX2_DTYPE_BF16 || dtype == LTX2_DTYPE_F32 || dtype == LTX2_DTYPE_FP8_E4M3FN, "dit_set_weight: bad dtype %d", dtype);
    c->weights[name] = Wt{ptr, dtype, (long)numel};
    c->resolved = false;
    c->prepared = false;        // the per-prompt caches (text K/V, the gathered AdaLN tables) were built from the weights registered before
    return LTX2_OK;
}

int64_t ltx2_dit_workspace_bytes_av(const ltx2_dit* c, int N, int S, int Na, int Sa, int per_token) {
    if (!c || N <= 0 || S <= 0 || (c->av && (Na <= 0 || Sa <= 0))) return -1;
    ltx2_dit tmp;            // carve() writes pointers; use a scratch context
    tmp.cfg = c->cfg;
    tmp.av = c->av;
    tmp.v2 = c->v2;
    tmp.gated = c->gated;
    tmp.fp8_compute = c->fp8_compute;
    tmp.m[0] = c->m[0];
    tmp.m[1] = c->m[1];
    return carve(&tmp, nullptr, N, S, Na, Sa, per_token);
}

int64_t ltx2_dit_workspace_bytes(const ltx2_dit* c, int N, int S, int per_token) {
    if (!c || c->av) return -1;
    return ltx2_dit_workspace_bytes_av(c, N, S, 0, 0, per_token);
}

int ltx2_dit_bind_workspace(ltx2_dit* c, void* ptr, int64_t bytes, int N, int S, int per_token) {
    LTX2_CHECK_ARG(c && ptr && N > 0 && S > 0, "dit_bind_workspace: bad argument");
    LTX2_CHECK_ARG(!c->av, "dit_bind_workspace: AudioVideo model, use ltx2_dit_bind_workspace_av");
    return bind(c, ptr, bytes, N, S, 0, 0, per_token);
}

int ltx2_dit_bind_workspace_av(ltx2_dit* c, void* ptr, int64_t bytes, int N, int S, int Na, int Sa, int per_token) {
    LTX2_CHECK_ARG(c && ptr && N > 0 && S > 0 && Na > 0 && Sa > 0, "dit_bind_workspace_av: bad argument");
    LTX2_CHECK_ARG(c->av, "dit_bind_workspace_av: VideoOnly model, use ltx2_dit_bind_workspace");
    return bind(c, ptr, bytes, N, S, Na, Sa, per_token);
}

int ltx2_dit_prepare(ltx2_dit* c, const float* context, int S, const float* rope_cos, const float* rope_sin,
                     void* stream) {
    LTX2_CHECK_ARG(c, "dit_prepare: null context");
    LTX2_CHECK_ARG(!c->av, "dit_prepare: AudioVideo model, use ltx2_dit_prepare_av");
    if (!c->ws) {
        ltx2_set_error("dit_prepare: no workspace bound");
        return LTX2_E_STATE;
    }
    TRY(resolve(c));
    TRY(prepare_modality(c, 0, context, S, rope_cos, rope_sin, nullptr, nullptr, (hipStream_t)stream));
    c->prepared = true;
    return LTX2_OK;
}

int ltx2_dit_prepare_av(ltx2_dit* c, const float* v_context, int S, const float* v_cos, const float* v_sin,
                        const float* v_cross_cos, const float* v_cross_sin, const float* a_context, int Sa,
                        const float* a_cos, const float* a_sin, const float* a_cross_cos, const float* a_cross_sin,
                        void* stream) {
    LTX2_CHECK_ARG(c, "dit_prepare_av: null context");
    LTX2_CHECK_ARG(c->av, "dit_prepare_av: VideoOnly model, use ltx2_dit_prepare");
    if (!c->ws) {
        ltx2_set_error("dit_prepare_av: no workspace bound");
        return LTX2_E_STATE;
    }
    TRY(resolve(c));
    TRY(prepare_modality(c, 0, v_context, S, v_cos, v_sin, v_cross_cos, v_cross_sin, (hipStream_t)stream));
    TRY(prepare_modality(c, 1, a_context, Sa, a_cos, a_sin, a_cross_cos, a_cross_sin, (hipStream_t)stream));
    c->prepared = true;
    return LTX2_OK;
}

int ltx2_dit_forward(ltx2_dit* c, const float* latent, const float* timesteps, int n_timesteps, const float* sigma,
                     float* velocity, void* stream) {
    LTX2_CHECK_ARG(latent && timesteps && velocity, "dit_forward: null argument");
    TRY(check_ready(c, "dit_forward", false));
    ModIn in[1] = {{latent, timesteps, n_timesteps, sigma ? sigma : timesteps, velocity}};
    return forward(c, in, (hipStream_t)stream);
}

int ltx2_dit_forward_av(ltx2_dit* c, const float* v_latent, const float* v_timesteps, int n_v_timesteps,
                        const float* v_sigma, const float* a_latent, const float* a_timesteps, int n_a_timesteps,
                        const float* a_sigma, float* v_velocity, float* a_velocity, void* stream) {
    LTX2_CHECK_ARG(v_latent && v_timesteps && v_sigma && a_latent && a_timesteps && a_sigma && v_velocity && a_velocity,
                   "dit_forward_av: null argument");
    TRY(check_ready(c, "dit_forward_av", true));
    ModIn in[2] = {{v_latent, v_timesteps, n_v_timesteps, v_sigma, v_velocity},
                   {a_latent, a_timesteps, n_a_timesteps, a_sigma, a_velocity}};
    return forward(c, in, (hipStream_t)stream);
}

int ltx2_dit_denoise_step(ltx2_dit* c, float* latent, const float* timesteps, int n_timesteps, const float* sigma_dev,
                          const float* mask, const float* clean, float sigma, float sigma_next, float* x0_out, void* stream) {
    LTX2_CHECK_ARG(latent && timesteps, "dit_denoise_step: null argument");
    TRY(check_ready(c, "dit_denoise_step", false));
    ModIn in[1] = {{latent, timesteps, n_timesteps, sigma_dev ? sigma_dev : timesteps, nullptr}};
    const StepIo io[1] = {{latent, mask, clean, x0_out}};
    return denoise_step(c, in, io, sigma, sigma_next, (hipStream_t)stream);
}

int ltx2_dit_denoise_step_av(ltx2_dit* c, float* v_latent, float* a_latent, const float* v_timesteps, int n_v_timesteps,
                             const float* a_timesteps, int n_a_timesteps, const float* sigma_dev, const float* v_mask,
                             const float* v_clean, const float* a_mask, const float* a_clean, float sigma,
                             float sigma_next, float* v_x0_out, float* a_x0_out, void* stream) {
    LTX2_CHECK_ARG(v_latent && a_latent && v_timesteps && a_timesteps && sigma_dev, "dit_denoise_step_av: null argument");
    TRY(check_ready(c, "dit_denoise_step_av", true));
    ModIn in[2] = {{v_latent, v_timesteps, n_v_timesteps, sigma_dev, nullptr},
                   {a_latent, a_timesteps, n_a_timesteps, sigma_dev, nullptr}};
    const StepIo io[2] = {{v_latent, v_mask, v_clean, v_x0_out}, {a_latent, a_mask, a_clean, a_x0_out}};
    return denoise_step(c, in, io, sigma, sigma_next, (hipStream_t)stream);
}

int ltx2_dit_graph_capture(ltx2_dit* c, float* latent, const float* host_sigmas, int n_steps, void* stream) {
    LTX2_CHECK_ARG(latent && host_sigmas && n_steps > 0 && n_steps < 64, "dit_graph_capture: bad argument");
    TRY(check_ready(c, "dit_graph_capture", false));
    hipStream_t st = (hipStream_t)stream;
    TRY(begin_capture(c, host_sigmas, n_steps, st));
    int rc = LTX2_OK;
    for (int i = 0; i < n_steps && rc == LTX2_OK; ++i) {
        ModIn in[1] = {{latent, c->sigmas_dev + i, 1, c->sigmas_dev + i, nullptr}};
        const StepIo io[1] = {{latent, nullptr, nullptr, nullptr}};
        rc = denoise_step(c, in, io, host_sigmas[i], host_sigmas[i + 1], st, i == 0);
    }
    return end_capture(c, rc, st);
}

// Conditioned loops (image-to-video: some tokens carry a denoise mask < 1): per step the timesteps are mask * sigma_i, formed on the device
// inside the captured step, and the Euler update blends x0 with the clean latent as ltx2_dit_denoise_step does.
namespace {
int cond_modality(ltx2_dit* c, int k, const float* mask, long n_mask, const float* clean, long n_clean, const float* sigma_i, ModIn& in, StepIo& io, hipStream_t st) {
    Mod& m = c->m[k];
    if (!mask) return LTX2_OK;           // this modality has no conditioning tokens: the uniform form
    LTX2_CHECK_ARG(clean, "dit_graph_capture_cond: a denoise mask needs the clean latent");
    // the replayed kernels read N mask elements and N * C clean-latent elements on every step: a buffer of another modality's length would be an
    // out-of-bounds device read, not an error
    LTX2_CHECK_ARG(n_mask == m.N && n_clean == (long)m.N * m.Cout, "dit_graph_capture_cond: modality %d has %d tokens x %d channels, got a mask of %ld and a clean latent of %ld elements", k, m.N, m.Cout, n_mask, n_clean);
    if (!c->per_token || !m.ts_tok) {
        ltx2_set_error("dit_graph_capture_cond: the workspace was not bound for per-token timesteps");
        return LTX2_E_STATE;
    }
    hipLaunchKernelGGL(mask_sigma_kernel, dim3((m.N + 255) / 256), dim3(256), 0, st, mask, sigma_i, m.ts_tok, m.N);
    LTX2_CHECK_LAUNCH("mask_sigma_kernel");
    in.ts = m.ts_tok;
    in.n_ts = m.N;
    io.mask = mask;
    io.clean = clean;
    return LTX2_OK;
}
}  // namespace

int ltx2_dit_graph_capture_cond(ltx2_dit* c, float* latent, const float* host_sigmas, int n_steps, const float* mask, int64_t n_mask, const float* clean,
                                int64_t n_clean, void* stream) {
    LTX2_CHECK_ARG(latent && host_sigmas && n_steps > 0 && n_steps < 64, "dit_graph_capture_cond: bad argument");
    TRY(check_ready(c, "dit_graph_capture_cond", false));
    hipStream_t st = (hipStream_t)stream;
    TRY(begin_capture(c, host_sigmas, n_steps, st));
    int rc = LTX2_OK;
    for (int i = 0; i < n_steps && rc == LTX2_OK; ++i) {
        ModIn in[1] = {{latent, c->sigmas_dev + i, 1, c->sigmas_dev + i, nullptr}};
        StepIo io[1] = {{latent, nullptr, nullptr, nullptr}};
        rc = cond_modality(c, 0, mask, n_mask, clean, n_clean, c->sigmas_dev + i, in[0], io[0], st);
        if (rc == LTX2_OK) rc = denoise_step(c, in, io, host_sigmas[i], host_sigmas[i + 1], st, i == 0);
    }
    return end_capture(c, rc, st);
}

int ltx2_dit_graph_capture_cond_av(ltx2_dit* c, float* v_latent, float* a_latent, const float* host_sigmas, int n_steps, const float* v_mask, int64_t n_v_mask,
                                   const float* v_clean, int64_t n_v_clean, const float* a_mask, int64_t n_a_mask, const float* a_clean, int64_t n_a_clean, void* stream) {
    LTX2_CHECK_ARG(v_latent && a_latent && host_sigmas && n_steps > 0 && n_steps < 64, "dit_graph_capture_cond_av: bad argument");
    TRY(check_ready(c, "dit_graph_capture_cond_av", true));
    hipStream_t st = (hipStream_t)stream;
    TRY(begin_capture(c, host_sigmas, n_steps, st));
    int rc = LTX2_OK;
    for (int i = 0; i < n_steps && rc == LTX2_OK; ++i) {
        const float* s = c->sigmas_dev + i;
        ModIn in[2] = {{v_latent, s, 1, s, nullptr}, {a_latent, s, 1, s, nullptr}};
        StepIo io[2] = {{v_latent, nullptr, nullptr, nullptr}, {a_latent, nullptr, nullptr, nullptr}};
        rc = cond_modality(c, 0, v_mask, n_v_mask, v_clean, n_v_clean, s, in[0], io[0], st);
        if (rc == LTX2_OK) rc = cond_modality(c, 1, a_mask, n_a_mask, a_clean, n_a_clean, s, in[1], io[1], st);
        if (rc == LTX2_OK) rc = denoise_step(c, in, io, host_sigmas[i], host_sigmas[i + 1], st, i == 0);
    }
    return end_capture(c, rc, st);
}

int ltx2_dit_graph_capture_av(ltx2_dit* c, float* v_latent, float* a_latent, const float* host_sigmas, int n_steps,
                              void* stream) {
    LTX2_CHECK_ARG(v_latent && a_latent && host_sigmas && n_steps > 0 && n_steps < 64, "dit_graph_capture_av: bad argument");
    TRY(check_ready(c, "dit_graph_capture_av", true));
    hipStream_t st = (hipStream_t)stream;
    TRY(begin_capture(c, host_sigmas, n_steps, st));
    int rc = LTX2_OK;
    for (int i = 0; i < n_steps && rc == LTX2_OK; ++i) {
        const float* s = c->sigmas_dev + i;
        ModIn in[2] = {{v_latent, s, 1, s, nullptr}, {a_latent, s, 1, s, nullptr}};
        const StepIo io[2] = {{v_latent, nullptr, nullptr, nullptr}, {a_latent, nullptr, nullptr, nullptr}};
        rc = denoise_step(c, in, io, host_sigmas[i], host_sigmas[i + 1], st, i == 0);
    }
    return end_capture(c, rc, st);
}

int ltx2_dit_profile_begin(ltx2_dit* c, int epilogue) {
    LTX2_CHECK_ARG(c && epilogue >= -1 && epilogue < EPI_COUNT, "dit_profile_begin: bad argument");
    c->prof_epi = epilogue;
    c->prof_used = 0;
    c->prof_flops = 0;
    g_prof_ctx = c;
    return LTX2_OK;
}

int ltx2_dit_profile_end(ltx2_dit* c, double* total_ms, int64_t* launches, double* flops) {
    LTX2_CHECK_ARG(c && total_ms && launches && flops, "dit_profile_end: null argument");
    g_prof_ctx = nullptr;
    c->prof_epi = -2;
    double tot = 0;
    for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
        if (hipEventSynchronize(c->prof_ev[i + 1]) != hipSuccess) {
            ltx2_set_error("dit_profile_end: event synchronize failed");
            return LTX2_E_HIP;
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]);
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)(c->prof_used / 2);
    *flops = c->prof_flops;
    return LTX2_OK;
}

int ltx2_dit_set_context_mask(ltx2_dit* c, int modality, const float* mask, int S, void* stream) {
    LTX2_CHECK_ARG(c && (modality == 0 || (modality == 1 && c->av)), "dit_set_context_mask: bad context / modality");
    Mod& m = c->m[modality];
    if (!c->ws) {
        ltx2_set_error("dit_set_context_mask: no workspace bound");
        return LTX2_E_STATE;
    }
    if (!mask) {
        m.has_kmask = false;
        return LTX2_OK;
    }
    LTX2_CHECK_ARG(S == m.S, "dit_set_context_mask: S=%d differs from the bound workspace S=%d", S, m.S);
    TRY(keymask_words_launch(mask, S, m.kmask, m.Spad / 64, (hipStream_t)stream));
    m.has_kmask = true;
    return LTX2_OK;
}

int ltx2_dit_set_option(ltx2_dit* c, const char* name, int value) {
    LTX2_CHECK_ARG(c && name, "dit_set_option: null argument");
    if (!strcmp(name, "fp8_compute")) {
        if (c->ws && (value != 0) != c->fp8_compute) {
            ltx2_set_error("dit_set_option: fp8_compute must be set before the workspace is bound");
            return LTX2_E_STATE;
        }
        c->fp8_compute = value != 0;
        return LTX2_OK;
    }
    if (!strcmp(name, "text_kv_ahead")) {       // 0: the video stream projects its sigma-modulated text K / V inline (round 4's schedule; same results bit for bit)
        c->text_kv_ahead = value != 0;
        return LTX2_OK;
    }
    if (!strcmp(name, "fold_norms")) {          // 0 / 1: see block_attention(); 0 restores round 5's norm pass in front of the text cross-attention
        LTX2_CHECK_ARG(value == 0 || value == 1, "dit_set_option: fold_norms is 0 or 1");
        c->fold_norms = value;
        return LTX2_OK;
    }
    if (!strcmp(name, "adaln_combine")) {       // 0: tables and embeddings reach every kernel separately (round 3's form; same results bit for bit)
        c->adaln_combine = value != 0;
        return LTX2_OK;
    }
    ltx2_set_error("dit_set_option: unknown option '%s'", name);
    return LTX2_E_INVALID;
}


int ltx2_dit_graph_launch(ltx2_dit* c, void* stream) {
    LTX2_CHECK_ARG(c, "dit_graph_launch: null context");
    if (!c->exec) {
        ltx2_set_error("dit_graph_launch: no captured graph");
        return LTX2_E_STATE;
    }
    if (hipGraphLaunch(c->exec, (hipStream_t)stream) != hipSuccess) {
        ltx2_set_error("dit_graph_launch: hipGraphLaunch failed");
        return LTX2_E_HIP;
    }
    return LTX2_OK;
}

}  // extern "C"
