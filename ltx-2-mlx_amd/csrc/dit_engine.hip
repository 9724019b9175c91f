// DiT engine: sequences the gfx950 kernels of one LTX-2 denoise step on a caller stream and owns
// the hipGraph of the distilled sampling loop.  Host-side only (no kernels here).
//
// Restates the control flow of the reference's LTXModel.__call__ / BasicTransformerBlock.__call__
// (LTX_2_MLX/model/transformer/model.py:776-881, transformer.py:191-238) with the step-invariant
// work (caption projection, cross-attention K/V, RoPE tables) hoisted into ltx2_dit_prepare.
#include <math.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ltx2hip.h"
#include "attention.h"
#include "gemm.h"
#include "rowops.h"

#define TRY(expr)                   \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != LTX2_OK) return rc_; \
    } while (0)

namespace {
struct Wt {
    const void* p;
    int dtype;
    long n;
};

struct LayerW {
    const bf16 *qkv_w, *o_w, *q2_w, *kv2_w, *o2_w, *ff1_w, *ff2_w;
    const float *qkv_b, *o_b, *q2_b, *kv2_b, *o2_b, *ff1_b, *ff2_b;
    const float *qn1, *kn1, *qn2, *kn2, *sst;
};

inline long align_up(long v, long a = 256) { return (v + a - 1) / a * a; }
}  // namespace

struct ltx2_dit {
    ltx2_dit_config cfg{};
    int D = 0;
    std::unordered_map<std::string, Wt> weights;
    std::vector<LayerW> layers;
    bool resolved = false;
    const bf16 *patch_w = nullptr, *t1_w = nullptr, *t2_w = nullptr, *ada_w = nullptr, *cap1_w = nullptr,
               *cap2_w = nullptr, *proj_w = nullptr;
    const float *patch_b = nullptr, *t1_b = nullptr, *t2_b = nullptr, *ada_b = nullptr, *cap1_b = nullptr,
                *cap2_b = nullptr, *proj_b = nullptr, *sst_out = nullptr;
    // workspace
    char* ws = nullptr;
    long ws_bytes = 0;
    int N = 0, S = 0, Npad = 0, Spad = 0, per_token = 0;
    float *x = nullptr, *sin_f = nullptr, *e1_f = nullptr, *e_f = nullptr, *emb = nullptr, *vel = nullptr,
          *x0 = nullptr, *cosb = nullptr, *sinb = nullptr, *sigmas_dev = nullptr;
    bf16 *lat = nullptr, *h = nullptr, *qkv = nullptr, *vt = nullptr, *att = nullptr, *ff = nullptr, *sin_b = nullptr,
         *e1_b = nullptr, *es_b = nullptr, *ctx_in = nullptr, *c1 = nullptr, *ctxp = nullptr, *kv2 = nullptr,
         *vt2 = nullptr;
    bool prepared = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // live HIP-event profiling of one GEMM kernel instantiation (bench.py roofline)
    int prof_epi = -2;                 // -2 off, -1 every GEMM, >= 0 one epilogue
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    double prof_flops = 0;
};

namespace {

// Carve the workspace; base == nullptr only computes the size.
long carve(ltx2_dit* c, char* base, int N, int S, int per_token) {
    const long D = c->D, L = c->cfg.num_layers, H = c->cfg.num_heads;
    const long Npad = align_up(N, 64), Spad = align_up(S, 64);
    const long Cctx = c->cfg.caption_channels > 0 ? c->cfg.caption_channels : D;
    const long T = per_token ? N : 1;
    long off = 0;
    auto take = [&](long bytes) {
        char* p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    c->x = (float*)take(4L * N * D);
    c->lat = (bf16*)take(2L * N * c->cfg.in_channels);
    c->h = (bf16*)take(2L * N * D);
    c->qkv = (bf16*)take(2L * N * 3 * D);
    c->vt = (bf16*)take(2L * H * 128 * Npad);
    c->att = (bf16*)take(2L * N * D);
    c->ff = (bf16*)take(2L * N * 4 * D);
    c->sin_f = (float*)take(4L * 256);
    c->e1_f = (float*)take(4L * D);
    c->e_f = (float*)take(4L * T * D);
    c->emb = (float*)take(4L * T * 6 * D);
    c->sin_b = (bf16*)take(per_token ? 2L * N * 256 : 0);
    c->e1_b = (bf16*)take(per_token ? 2L * N * D : 0);
    c->es_b = (bf16*)take(per_token ? 2L * N * D : 0);
    c->vel = (float*)take(4L * N * c->cfg.out_channels);
    c->x0 = (float*)take(4L * N * c->cfg.out_channels);
    c->cosb = (float*)take(4L * N * (D / 2));
    c->sinb = (float*)take(4L * N * (D / 2));
    c->sigmas_dev = (float*)take(4L * 64);
    c->ctx_in = (bf16*)take(2L * S * Cctx);
    c->c1 = (bf16*)take(2L * S * D);
    c->ctxp = (bf16*)take(2L * S * D);
    c->kv2 = (bf16*)take(2L * L * S * 2 * D);
    c->vt2 = (bf16*)take(2L * L * H * 128 * Spad);
    return off;
}

const void* find(ltx2_dit* c, const std::string& name, int dtype, long numel, bool required = true) {
    auto it = c->weights.find(name);
    if (it == c->weights.end()) {
        if (required) ltx2_set_error("dit: missing weight '%s'", name.c_str());
        return nullptr;
    }
    if (it->second.dtype != dtype || it->second.n != numel) {
        ltx2_set_error("dit: weight '%s' has dtype %d / numel %ld, expected dtype %d / numel %ld", name.c_str(),
                       it->second.dtype, it->second.n, dtype, numel);
        return nullptr;
    }
    return it->second.p;
}

int resolve(ltx2_dit* c) {
    if (c->resolved) return LTX2_OK;
    const long D = c->D;
#define GETW(dst, name, rows, cols)                                                         \
    do {                                                                                    \
        dst = (const bf16*)find(c, std::string(name) + ".weight", LTX2_DTYPE_BF16, (long)(rows) * (cols)); \
        if (!dst) return LTX2_E_STATE;                                                      \
    } while (0)
#define GETB(dst, name, n)                                                                  \
    do {                                                                                    \
        dst = (const float*)find(c, std::string(name) + ".bias", LTX2_DTYPE_F32, (long)(n)); \
        if (!dst) return LTX2_E_STATE;                                                      \
    } while (0)
#define GETF(dst, name, n)                                                                  \
    do {                                                                                    \
        dst = (const float*)find(c, std::string(name), LTX2_DTYPE_F32, (long)(n));          \
        if (!dst) return LTX2_E_STATE;                                                      \
    } while (0)
    GETW(c->patch_w, "patchify_proj", D, c->cfg.in_channels);
    GETB(c->patch_b, "patchify_proj", D);
    GETW(c->t1_w, "adaln_single.emb.timestep_embedder.linear_1", D, 256);
    GETB(c->t1_b, "adaln_single.emb.timestep_embedder.linear_1", D);
    GETW(c->t2_w, "adaln_single.emb.timestep_embedder.linear_2", D, D);
    GETB(c->t2_b, "adaln_single.emb.timestep_embedder.linear_2", D);
    GETW(c->ada_w, "adaln_single.linear", 6 * D, D);
    GETB(c->ada_b, "adaln_single.linear", 6 * D);
    if (c->cfg.caption_channels > 0) {
        GETW(c->cap1_w, "caption_projection.linear_1", D, c->cfg.caption_channels);
        GETB(c->cap1_b, "caption_projection.linear_1", D);
        GETW(c->cap2_w, "caption_projection.linear_2", D, D);
        GETB(c->cap2_b, "caption_projection.linear_2", D);
    }
    GETF(c->sst_out, "scale_shift_table", 2 * D);
    GETW(c->proj_w, "proj_out", c->cfg.out_channels, D);
    GETB(c->proj_b, "proj_out", c->cfg.out_channels);
    c->layers.resize(c->cfg.num_layers);
    for (int i = 0; i < c->cfg.num_layers; ++i) {
        LayerW& w = c->layers[i];
        const std::string p = "transformer_blocks." + std::to_string(i);
        GETW(w.qkv_w, p + ".attn1.to_qkv", 3 * D, D);
        GETB(w.qkv_b, p + ".attn1.to_qkv", 3 * D);
        GETW(w.o_w, p + ".attn1.to_out.0", D, D);
        GETB(w.o_b, p + ".attn1.to_out.0", D);
        GETF(w.qn1, p + ".attn1.q_norm.weight", D);
        GETF(w.kn1, p + ".attn1.k_norm.weight", D);
        GETW(w.q2_w, p + ".attn2.to_q", D, D);
        GETB(w.q2_b, p + ".attn2.to_q", D);
        GETW(w.kv2_w, p + ".attn2.to_kv", 2 * D, D);
        GETB(w.kv2_b, p + ".attn2.to_kv", 2 * D);
        GETW(w.o2_w, p + ".attn2.to_out.0", D, D);
        GETB(w.o2_b, p + ".attn2.to_out.0", D);
        GETF(w.qn2, p + ".attn2.q_norm.weight", D);
        GETF(w.kn2, p + ".attn2.k_norm.weight", D);
        GETW(w.ff1_w, p + ".ff.net.0.proj", 4 * D, D);
        GETB(w.ff1_b, p + ".ff.net.0.proj", 4 * D);
        GETW(w.ff2_w, p + ".ff.net.2", D, 4 * D);
        GETB(w.ff2_b, p + ".ff.net.2", D);
        GETF(w.sst, p + ".scale_shift_table", 6 * D);
    }
#undef GETW
#undef GETB
#undef GETF
    c->resolved = true;
    return LTX2_OK;
}

thread_local ltx2_dit* g_prof_ctx = nullptr;

int dense(const bf16* A, long lda, const bf16* W, const float* bias, void* out, long ldo, int M, int N, int K, int epi,
          hipStream_t st, const float* gate = nullptr, long gate_stride = 0, const float* gate_table = nullptr) {
    GemmParams p{};
    p.A = A;
    p.lda = lda;
    p.W = W;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate = gate;
    p.gate_stride = gate_stride;
    p.gate_table = gate_table;
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

__global__ void silu_cast_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = f2bf(silu_f(in[i]));
}

int forward(ltx2_dit* c, const float* latent, const float* timesteps, int n_ts, float* velocity, hipStream_t st) {
    const int N = c->N, D = c->D, H = c->cfg.num_heads, Cin = c->cfg.in_channels;
    const float eps = c->cfg.norm_eps;
    LTX2_CHECK_ARG(n_ts == 1 || n_ts == N, "dit_forward: n_timesteps=%d must be 1 or N=%d", n_ts, N);
    LTX2_CHECK_ARG(n_ts == 1 || c->per_token, "dit_forward: workspace was not sized for per-token timesteps");
    const float attn_scale = 1.0f / sqrtf((float)c->cfg.head_dim);

    // patchify_proj (model.py:242) -> fp32 residual stream
    TRY(cast_f32_bf16_launch(latent, c->lat, (long)N * Cin, st));
    TRY(dense(c->lat, Cin, c->patch_w, c->patch_b, c->x, D, N, D, Cin, EPI_F32, st));

    // AdaLN-single (model.py:113-140; timestep_embedding.py:187-202)
    long es = 0, ee = 0;   // row strides of emb / e
    if (n_ts == 1) {
        TRY(timestep_sinusoid_launch(timesteps, 0, 0.f, c->cfg.timestep_scale, 1, 256, c->sin_f, nullptr, st));
        TRY(gemv_launch(c->sin_f, 256, c->t1_w, c->t1_b, c->e1_f, D, 1, D, 256, 0, 1, st));
        TRY(gemv_launch(c->e1_f, D, c->t2_w, c->t2_b, c->e_f, D, 1, D, D, 0, 0, st));
        TRY(gemv_launch(c->e_f, D, c->ada_w, c->ada_b, c->emb, 6 * D, 1, 6 * D, D, 1, 0, st));
    } else {
        TRY(timestep_sinusoid_launch(timesteps, 1, 0.f, c->cfg.timestep_scale, N, 256, nullptr, c->sin_b, st));
        TRY(dense(c->sin_b, 256, c->t1_w, c->t1_b, c->e1_b, D, N, D, 256, EPI_SILU_BF16, st));
        TRY(dense(c->e1_b, D, c->t2_w, c->t2_b, c->e_f, D, N, D, D, EPI_F32, st));
        hipLaunchKernelGGL(silu_cast_kernel, dim3(2048), dim3(256), 0, st, c->e_f, c->es_b, (long)N * D);
        LTX2_CHECK_LAUNCH("silu_cast_kernel");
        TRY(dense(c->es_b, D, c->ada_w, c->ada_b, c->emb, 6 * D, N, 6 * D, D, EPI_F32, st));
        es = 6L * D;
        ee = D;
    }
    const float* emb = c->emb;

    for (int l = 0; l < c->cfg.num_layers; ++l) {
        const LayerW& w = c->layers[l];
        // self-attention: AdaLN rows (shift, scale, gate) = sst[0:3] + emb[0:3]  (transformer.py:207-214)
        TRY(norm_mod_launch(c->x, D, c->h, D, N, D, eps, 0, w.sst + D, w.sst, emb + D, emb, es, st));
        TRY(dense(c->h, D, w.qkv_w, w.qkv_b, c->qkv, 3 * D, N, 3 * D, D, EPI_BF16, st));
        {
            const int offs[2] = {0, D};
            const float* wts[2] = {w.qn1, w.kn1};
            TRY(qknorm_rope_launch(c->qkv, 3 * D, N, D, c->cfg.head_dim, 2, offs, wts, eps, c->cosb, c->sinb, st));
        }
        TRY(vt_transpose_launch(c->qkv + 2 * D, 3 * D, c->vt, N, c->Npad, H, st));
        {
            AttnParams a{};
            a.Q = c->qkv;
            a.ldq = 3 * D;
            a.K = c->qkv + D;
            a.ldk = 3 * D;
            a.VT = c->vt;
            a.vt_head_stride = 128L * c->Npad;
            a.O = c->att;
            a.ldo = D;
            a.Nq = N;
            a.Nkv = N;
            a.Npad = c->Npad;
            a.H = H;
            a.scale_log2e = attn_scale * 1.4426950408889634f;
            TRY(attn_launch(a, st));
        }
        TRY(dense(c->att, D, w.o_w, w.o_b, c->x, D, N, D, D, EPI_RESID_GATE_F32, st, emb + 2 * D, es, w.sst + 2 * D));

        // text cross-attention: plain RMSNorm on x, no RoPE, no mask, no gate (transformer.py:217-226)
        TRY(norm_mod_launch(c->x, D, c->h, D, N, D, eps, 0, nullptr, nullptr, nullptr, nullptr, 0, st));
        TRY(dense(c->h, D, w.q2_w, w.q2_b, c->qkv, D, N, D, D, EPI_BF16, st));
        {
            const int offs[1] = {0};
            const float* wts[1] = {w.qn2};
            TRY(qknorm_rope_launch(c->qkv, D, N, D, c->cfg.head_dim, 1, offs, wts, eps, nullptr, nullptr, st));
        }
        {
            AttnParams a{};
            a.Q = c->qkv;
            a.ldq = D;
            a.K = c->kv2 + (long)l * c->S * 2 * D;
            a.ldk = 2 * D;
            a.VT = c->vt2 + (long)l * H * 128 * c->Spad;
            a.vt_head_stride = 128L * c->Spad;
            a.O = c->att;
            a.ldo = D;
            a.Nq = N;
            a.Nkv = c->S;
            a.Npad = c->Spad;
            a.H = H;
            a.scale_log2e = attn_scale * 1.4426950408889634f;
            TRY(attn_launch(a, st));
        }
        TRY(dense(c->att, D, w.o2_w, w.o2_b, c->x, D, N, D, D, EPI_RESID_GATE_F32, st));

        // feed-forward: AdaLN rows 3..5 (transformer.py:229-236); Linear -> GELU(tanh) -> Linear, ungated
        TRY(norm_mod_launch(c->x, D, c->h, D, N, D, eps, 0, w.sst + 4 * D, w.sst + 3 * D, emb + 4 * D, emb + 3 * D, es, st));
        TRY(dense(c->h, D, w.ff1_w, w.ff1_b, c->ff, 4 * D, N, 4 * D, D, EPI_GELU_BF16, st));
        TRY(dense(c->ff, 4 * D, w.ff2_w, w.ff2_b, c->x, D, N, D, 4 * D, EPI_RESID_GATE_F32, st, emb + 5 * D, es, w.sst + 5 * D));
    }

    // output head (model.py:744-758): LayerNorm(no affine) * (1 + scale) + shift, rows (shift, scale)
    TRY(norm_mod_launch(c->x, D, c->h, D, N, D, eps, 1, c->sst_out + D, c->sst_out, c->e_f, c->e_f, ee, st));
    TRY(dense(c->h, D, c->proj_w, c->proj_b, velocity, c->cfg.out_channels, N, c->cfg.out_channels, D, EPI_F32, st));
    return LTX2_OK;
}

int denoise_step(ltx2_dit* c, float* latent, const float* timesteps, int n_ts, const float* mask, const float* clean,
                 float sigma, float sigma_next, float* x0_out, hipStream_t st) {
    TRY(forward(c, latent, timesteps, n_ts, c->vel, st));
    float* x0 = x0_out ? x0_out : c->x0;
    const int C = c->cfg.out_channels;
    TRY(x0_from_velocity_launch(latent, c->vel, timesteps, n_ts == 1 ? 0 : 1, 0.f, x0, c->N, C, st));
    TRY(euler_step_launch(latent, x0, mask, clean, sigma, sigma_next, latent, c->N, C, st));
    return LTX2_OK;
}

}  // namespace

extern "C" {

int ltx2_dit_create(const ltx2_dit_config* cfg, ltx2_dit** out) {
    LTX2_CHECK_ARG(cfg && out, "dit_create: null argument");
    LTX2_CHECK_ARG(cfg->head_dim == 128, "dit_create: head_dim=%d, only 128 is implemented", cfg->head_dim);
    LTX2_CHECK_ARG(cfg->num_layers > 0 && cfg->num_heads > 0, "dit_create: bad layer/head count");
    LTX2_CHECK_ARG(cfg->in_channels % 64 == 0, "dit_create: in_channels must be a multiple of 64");
    LTX2_CHECK_ARG(cfg->caption_channels % 64 == 0, "dit_create: caption_channels must be a multiple of 64");
    ltx2_dit* c = new ltx2_dit();
    c->cfg = *cfg;
    c->D = cfg->num_heads * cfg->head_dim;
    *out = c;
    return LTX2_OK;
}

void ltx2_dit_destroy(ltx2_dit* c) {
    if (!c) return;
    if (c->exec) (void)hipGraphExecDestroy(c->exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    delete c;
}

int ltx2_dit_set_weight(ltx2_dit* c, const char* name, const void* ptr, int dtype, int64_t numel) {
    LTX2_CHECK_ARG(c && name && ptr, "dit_set_weight: null argument");
    LTX2_CHECK_ARG(dtype == LTX2_DTYPE_BF16 || dtype == LTX2_DTYPE_F32, "dit_set_weight: bad dtype %d", dtype);
    c->weights[name] = Wt{ptr, dtype, (long)numel};
    c->resolved = false;
    return LTX2_OK;
}

int64_t ltx2_dit_workspace_bytes(const ltx2_dit* c, int N, int S, int per_token) {
    if (!c || N <= 0 || S <= 0) return -1;
    ltx2_dit tmp;            // carve() writes pointers; use a scratch context
    tmp.cfg = c->cfg;
    tmp.D = c->D;
    return carve(&tmp, nullptr, N, S, per_token);
}

int ltx2_dit_bind_workspace(ltx2_dit* c, void* ptr, int64_t bytes, int N, int S, int per_token) {
    LTX2_CHECK_ARG(c && ptr && N > 0 && S > 0, "dit_bind_workspace: bad argument");
    LTX2_CHECK_ARG(((uintptr_t)ptr & 255) == 0, "dit_bind_workspace: pointer must be 256-byte aligned");
    const long need = carve(c, (char*)ptr, N, S, per_token);
    if (bytes < need) {
        ltx2_set_error("dit_bind_workspace: %ld bytes given, %ld needed", (long)bytes, need);
        return LTX2_E_STATE;
    }
    c->ws = (char*)ptr;
    c->ws_bytes = bytes;
    c->N = N;
    c->S = S;
    c->Npad = (int)align_up(N, 64);
    c->Spad = (int)align_up(S, 64);
    c->per_token = per_token;
    c->prepared = false;
    return LTX2_OK;
}

int ltx2_dit_prepare(ltx2_dit* c, const float* context, int S, const float* rope_cos, const float* rope_sin,
                     void* stream) {
    LTX2_CHECK_ARG(c && context && rope_cos && rope_sin, "dit_prepare: null argument");
    if (!c->ws) {
        ltx2_set_error("dit_prepare: no workspace bound");
        return LTX2_E_STATE;
    }
    LTX2_CHECK_ARG(S == c->S, "dit_prepare: S=%d differs from the bound workspace S=%d", S, c->S);
    TRY(resolve(c));
    hipStream_t st = (hipStream_t)stream;
    const int D = c->D, H = c->cfg.num_heads;
    const int Cctx = c->cfg.caption_channels > 0 ? c->cfg.caption_channels : D;
    const float eps = c->cfg.norm_eps;
    if (hipMemcpyAsync(c->cosb, rope_cos, 4L * c->N * (D / 2), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(c->sinb, rope_sin, 4L * c->N * (D / 2), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        ltx2_set_error("dit_prepare: RoPE table copy failed");
        return LTX2_E_HIP;
    }
    TRY(cast_f32_bf16_launch(context, c->ctx_in, (long)S * Cctx, st));
    const bf16* ctx = c->ctx_in;
    if (c->cfg.caption_channels > 0) {   // PixArtAlphaTextProjection (model.py:52-56)
        TRY(dense(c->ctx_in, Cctx, c->cap1_w, c->cap1_b, c->c1, D, S, D, Cctx, EPI_GELU_BF16, st));
        TRY(dense(c->c1, D, c->cap2_w, c->cap2_b, c->ctxp, D, S, D, D, EPI_BF16, st));
        ctx = c->ctxp;
    }
    for (int l = 0; l < c->cfg.num_layers; ++l) {
        const LayerW& w = c->layers[l];
        bf16* kv = c->kv2 + (long)l * S * 2 * D;
        TRY(dense(ctx, D, w.kv2_w, w.kv2_b, kv, 2 * D, S, 2 * D, D, EPI_BF16, st));
        const int offs[1] = {0};
        const float* wts[1] = {w.kn2};
        TRY(qknorm_rope_launch(kv, 2 * D, S, D, c->cfg.head_dim, 1, offs, wts, eps, nullptr, nullptr, st));
        TRY(vt_transpose_launch(kv + D, 2 * D, c->vt2 + (long)l * H * 128 * c->Spad, S, c->Spad, H, st));
    }
    c->prepared = true;
    return LTX2_OK;
}

int ltx2_dit_forward(ltx2_dit* c, const float* latent, const float* timesteps, int n_timesteps, float* velocity,
                     void* stream) {
    LTX2_CHECK_ARG(c && latent && timesteps && velocity, "dit_forward: null argument");
    if (!c->prepared) {
        ltx2_set_error("dit_forward: ltx2_dit_prepare has not been called");
        return LTX2_E_STATE;
    }
    return forward(c, latent, timesteps, n_timesteps, velocity, (hipStream_t)stream);
}

int ltx2_dit_denoise_step(ltx2_dit* c, float* latent, const float* timesteps, int n_timesteps, const float* mask,
                          const float* clean, float sigma, float sigma_next, float* x0_out, void* stream) {
    LTX2_CHECK_ARG(c && latent && timesteps, "dit_denoise_step: null argument");
    if (!c->prepared) {
        ltx2_set_error("dit_denoise_step: ltx2_dit_prepare has not been called");
        return LTX2_E_STATE;
    }
    return denoise_step(c, latent, timesteps, n_timesteps, mask, clean, sigma, sigma_next, x0_out, (hipStream_t)stream);
}

int ltx2_dit_graph_capture(ltx2_dit* c, float* latent, const float* host_sigmas, int n_steps, void* stream) {
    LTX2_CHECK_ARG(c && latent && host_sigmas && n_steps > 0 && n_steps < 64, "dit_graph_capture: bad argument");
    if (!c->prepared) {
        ltx2_set_error("dit_graph_capture: ltx2_dit_prepare has not been called");
        return LTX2_E_STATE;
    }
    hipStream_t st = (hipStream_t)stream;
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
    int rc = LTX2_OK;
    for (int i = 0; i < n_steps && rc == LTX2_OK; ++i)
        rc = denoise_step(c, latent, c->sigmas_dev + i, 1, nullptr, nullptr, host_sigmas[i], host_sigmas[i + 1], nullptr, st);
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
