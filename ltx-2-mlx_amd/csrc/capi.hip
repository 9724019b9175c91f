// C ABI (include/ltx2hip.h): error channel + thin extern "C" wrappers over the kernel launchers.
// The DiT / VAE engine entry points live in dit_engine.hip / vae_engine.hip.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ltx2hip.h"
#include "attention.h"
#include "gemm.h"
#include "rowops.h"

static thread_local char g_err[512] = "";

void ltx2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* ltx2_last_error(void) { return g_err; }
void ltx2_clear_error(void) { g_err[0] = 0; }
int ltx2_abi_version(void) { return LTX2_ABI_VERSION; }

int ltx2_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M,
                   int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table,
                   const void* res, int64_t ldres, void* stream) {
    LTX2_CHECK_ARG(epilogue >= 0 && epilogue <= LTX2_EPI_ADD_BF16, "gemm: epilogue %d out of range", epilogue);
    LTX2_CHECK_ARG(epilogue != LTX2_EPI_ADD_BF16 || res, "gemm: epilogue ADD_BF16 needs res");
    GemmParams p{};
    p.A = (const bf16*)A;
    p.lda = lda;
    p.W = (const bf16*)W;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate = gate;
    p.gate_stride = gate_stride;
    p.gate_table = gate_table;
    p.res = (const bf16*)res;
    p.ldres = ldres;
    return gemm_launch(p, epilogue, false, (hipStream_t)stream);
}

int ltx2_adaln_rmsnorm2(const float* x, int64_t ldx, void* out0, void* out1, int64_t ldo, int rows, int D, float eps, const float* scale0,
                        const float* shift0, const float* scale1, const float* shift1, void* stream) {
    LTX2_CHECK_ARG(x && out0 && out1, "adaln_rmsnorm2: null operand");
    const float* sct[2] = {scale0, scale1};
    const float* sht[2] = {shift0, shift1};
    const float* none[2] = {nullptr, nullptr};
    return norm_mod2_launch(x, ldx, (bf16*)out0, (bf16*)out1, ldo, rows, D, eps, sct, sht, none, none, (hipStream_t)stream);
}

int ltx2_gemm_bf16_rowss(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K, float* rowss,
                         int* written, void* stream) {
    LTX2_CHECK_ARG(A && W && out && rowss && written, "gemm_bf16_rowss: null argument");
    GemmParams p{};
    p.A = (const bf16*)A;
    p.lda = lda;
    p.W = (const bf16*)W;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    *written = gemm_rowss_supported(p, EPI_BF16) ? 1 : 0;
    if (*written) p.rowss = rowss;
    return gemm_launch(p, EPI_BF16, false, (hipStream_t)stream);
}

int ltx2_gemm_bf16_fold(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                        const float* gate_table, void* shadow, int64_t ld_shadow, const float* shadow_scale, float* shadow_ss, int64_t ld_ss,
                        const float* rf_parts, int64_t rf_ld, int rf_nparts, int rf_dim, float rf_eps, int* supported, void* stream) {
    LTX2_CHECK_ARG(A && W && out && supported, "gemm_bf16_fold: null argument");
    LTX2_CHECK_ARG(epilogue == LTX2_EPI_BF16 || epilogue == LTX2_EPI_GELU_BF16 || epilogue == LTX2_EPI_RESID_GATE_F32, "gemm_bf16_fold: epilogue %d (BF16, GELU_BF16 or RESID_GATE_F32)", epilogue);
    GemmParams p{};
    p.A = (const bf16*)A;
    p.lda = lda;
    p.W = (const bf16*)W;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate_table = gate_table;
    p.shadow = (bf16*)shadow;
    p.ld_shadow = ld_shadow;
    p.shadow_scale = shadow_scale;
    p.shadow_ss = shadow_ss;
    p.ld_ss = ld_ss;
    p.rf_parts = rf_parts;
    p.rf_ld = rf_ld;
    p.rf_nparts = rf_nparts;
    p.rf_dim = rf_dim;
    p.rf_eps = rf_eps;
    *supported = ((p.shadow || p.rf_parts) && gemm_fold_supported(p, epilogue)) ? 1 : 0;
    if (!*supported) return LTX2_OK;
    return gemm_launch(p, epilogue, false, (hipStream_t)stream);
}

int ltx2_flash_attn_rowscale(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                             int H, int head_dim, float scale, const float* q_ss, int q_ss_ld, int q_norm_dim, float q_eps, void* stream) {
    LTX2_CHECK_ARG(Q && K && VT && out && q_ss, "flash_attn_rowscale: null operand");
    LTX2_CHECK_ARG(head_dim == 128, "flash_attn_rowscale: head_dim=%d, only 128 is implemented", head_dim);
    AttnParams a{};
    a.Q = (const bf16*)Q;
    a.ldq = ldq;
    a.K = (const bf16*)K;
    a.ldk = ldk;
    a.VT = (const bf16*)VT;
    a.vt_head_stride = (long)head_dim * Npad;
    a.head_dim = head_dim;
    a.O = (bf16*)out;
    a.ldo = ldo;
    a.Nq = Nq;
    a.Nkv = Nkv;
    a.Npad = Npad;
    a.H = H;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.q_ss = q_ss;
    a.q_ss_ld = q_ss_ld;
    a.q_norm_dim = q_norm_dim;
    a.q_eps = q_eps;
    return attn_launch(a, (hipStream_t)stream);
}

int ltx2_flash_attn_keymask(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                            int H, int head_dim, float scale, const float* mask, void* words, void* stream) {
    LTX2_CHECK_ARG(Q && K && VT && out && mask && words, "flash_attn_keymask: null operand");
    if (const int rc = keymask_words_launch(mask, Nkv, (unsigned long long*)words, Npad / 64, (hipStream_t)stream)) return rc;
    AttnParams a{};
    a.Q = (const bf16*)Q;
    a.ldq = ldq;
    a.K = (const bf16*)K;
    a.ldk = ldk;
    a.VT = (const bf16*)VT;
    a.vt_head_stride = (long)head_dim * Npad;
    a.head_dim = head_dim;
    a.O = (bf16*)out;
    a.ldo = ldo;
    a.Nq = Nq;
    a.Nkv = Nkv;
    a.Npad = Npad;
    a.H = H;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.kmask = (const unsigned long long*)words;
    return attn_launch(a, (hipStream_t)stream);
}

int ltx2_gemm_route(int M, int N, int K, int epilogue, int weights, int has_vt) {
    // host logic only: addresses are never dereferenced (16-byte aligned dummies satisfy the alignment checks)
    static const long dummy[2] = {0, 0};
    GemmParams p{};
    p.M = M;
    p.N = N;
    p.K = K;
    p.lda = K;
    p.ldo = N;
    p.out = (void*)dummy;
    if (weights == 2) {
        p.A8 = (const unsigned char*)dummy;
        p.ascale = (const float*)dummy;
    } else {
        p.A = (const bf16*)dummy;
    }
    if (weights >= 1) {
        p.W8 = (const unsigned char*)dummy;
        p.wscale = (const float*)dummy;
    } else {
        p.W = (const bf16*)dummy;
    }
    const int r = gemm_route(p, epilogue, false);
    if (!has_vt || r < 0) return r;
    p.vt = (bf16*)dummy;
    p.vt_col0 = 2 * (N / 3);
    p.vt_hd = (N / 3) % 128 == 0 ? 128 : 64;
    p.vt_npad = (M + 63) / 64 * 64;
    return r | (gemm_vt_fused(p, epilogue) ? 0x100 : 0);
}

int ltx2_quantize_rows_fp8(const void* x, int64_t ldx, int rows, int K, void* codes, int64_t ldo, float* scale, void* stream) {
    LTX2_CHECK_ARG(x && codes && scale, "quantize_rows_fp8: null argument");
    return quant_rows_fp8_launch((const bf16*)x, ldx, rows, K, (unsigned char*)codes, ldo, scale, (hipStream_t)stream);
}

int ltx2_gemm_fp8(const void* A8, int64_t lda, const float* ascale, const void* W8, const float* wscale, const float* bias, void* out,
                  int64_t ldo, int M, int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table,
                  void* stream) {
    LTX2_CHECK_ARG(A8 && ascale && W8 && wscale && out, "gemm_fp8: null operand");
    GemmParams p{};
    p.A8 = (const unsigned char*)A8;
    p.ascale = ascale;
    p.lda = lda;
    p.W8 = (const unsigned char*)W8;
    p.wscale = wscale;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate = gate;
    p.gate_stride = gate_stride;
    p.gate_table = gate_table;
    return gemm_launch(p, epilogue, false, (hipStream_t)stream);
}

int ltx2_gemm_qkv_vt(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                     void* vt, int vt_col0, int Npad, int head_dim, int* fused, void* stream) {
    LTX2_CHECK_ARG(A && W && out && vt, "gemm_qkv_vt: null operand");
    LTX2_CHECK_ARG(head_dim == 64 || head_dim == 128, "gemm_qkv_vt: head_dim=%d, only 128 and 64 are implemented", head_dim);
    LTX2_CHECK_ARG(vt_col0 > 0 && vt_col0 < N && (N - vt_col0) % head_dim == 0 && Npad % 64 == 0 && Npad >= M, "gemm_qkv_vt: bad V column range / Npad");
    GemmParams p{};
    p.A = (const bf16*)A;
    p.lda = lda;
    p.W = (const bf16*)W;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.vt = (bf16*)vt;
    p.vt_col0 = vt_col0;
    p.vt_npad = Npad;
    p.vt_hd = head_dim;
    p.vt_head_stride = (long)head_dim * Npad;
    const bool f = gemm_vt_fused(p, EPI_BF16);
    if (fused) *fused = f ? 1 : 0;
    if (!f) p.vt = nullptr;
    int rc = gemm_launch(p, EPI_BF16, false, (hipStream_t)stream);
    if (rc != LTX2_OK || f) return rc;
    return vt_transpose_launch((const bf16*)out + vt_col0, ldo, (bf16*)vt, M, Npad, (N - vt_col0) / head_dim, (hipStream_t)stream, head_dim);
}

int ltx2_gemm_fp8_qkv_vt(const void* A8, int64_t lda, const float* ascale, const void* W8, const float* wscale, const float* bias, void* out,
                         int64_t ldo, int M, int N, int K, void* vt, int vt_col0, int Npad, int head_dim, int* fused, void* stream) {
    LTX2_CHECK_ARG(A8 && ascale && W8 && wscale && out && vt, "gemm_fp8_qkv_vt: null operand");
    LTX2_CHECK_ARG(head_dim == 64 || head_dim == 128, "gemm_fp8_qkv_vt: head_dim=%d, only 128 and 64 are implemented", head_dim);
    LTX2_CHECK_ARG(vt_col0 > 0 && vt_col0 < N && (N - vt_col0) % head_dim == 0 && Npad % 64 == 0 && Npad >= M, "gemm_fp8_qkv_vt: bad V column range / Npad");
    GemmParams p{};
    p.A8 = (const unsigned char*)A8;
    p.ascale = ascale;
    p.lda = lda;
    p.W8 = (const unsigned char*)W8;
    p.wscale = wscale;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.vt = (bf16*)vt;
    p.vt_col0 = vt_col0;
    p.vt_npad = Npad;
    p.vt_hd = head_dim;
    p.vt_head_stride = (long)head_dim * Npad;
    const bool f = gemm_vt_fused(p, EPI_BF16);
    if (fused) *fused = f ? 1 : 0;
    if (!f) p.vt = nullptr;
    int rc = gemm_launch(p, EPI_BF16, false, (hipStream_t)stream);
    if (rc != LTX2_OK || f) return rc;
    return vt_transpose_launch((const bf16*)out + vt_col0, ldo, (bf16*)vt, M, Npad, (N - vt_col0) / head_dim, (hipStream_t)stream, head_dim);
}

int ltx2_gemm_w8a16(const void* A, int64_t lda, const void* W8, const float* wscale, const float* bias, void* out, int64_t ldo, int M,
                    int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table, void* stream) {
    LTX2_CHECK_ARG(A && W8 && wscale && out, "gemm_w8a16: null operand");
    GemmParams p{};
    p.A = (const bf16*)A;
    p.lda = lda;
    p.W8 = (const unsigned char*)W8;
    p.wscale = wscale;
    p.bias = bias;
    p.out = out;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.gate = gate;
    p.gate_stride = gate_stride;
    p.gate_table = gate_table;
    return gemm_launch(p, epilogue, false, (hipStream_t)stream);
}

int ltx2_gemv_f32(const float* a, int64_t lda, const void* W, const float* bias, float* out, int64_t ldo, int M,
                  int N, int K, int act_in, int act_out, void* stream) {
    LTX2_CHECK_ARG(a && W && out, "gemv: null operand");
    return gemv_launch(a, lda, (const bf16*)W, bias, out, ldo, M, N, K, act_in, act_out, (hipStream_t)stream);
}

int ltx2_adaln_rmsnorm(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int D, float eps,
                       int layer_norm, const float* scale_tab, const float* shift_tab, const float* scale_emb,
                       const float* shift_emb, int64_t emb_stride, void* stream) {
    LTX2_CHECK_ARG(x && out, "adaln_rmsnorm: null operand");
    return norm_mod_launch(x, ldx, (bf16*)out, ldo, rows, D, eps, layer_norm, scale_tab, shift_tab, scale_emb, shift_emb,
                           emb_stride, (hipStream_t)stream);
}

int ltx2_adaln_rmsnorm_fp8(const float* x, int64_t ldx, void* out_bf16, int64_t ldo, void* codes, int64_t ldq, float* scale, int rows, int D,
                           float eps, int layer_norm, const float* scale_tab, const float* shift_tab, const float* scale_emb,
                           const float* shift_emb, int64_t emb_stride, void* stream) {
    LTX2_CHECK_ARG(x && codes && scale, "adaln_rmsnorm_fp8: null operand");
    return norm_mod_launch(x, ldx, (bf16*)out_bf16, ldo, rows, D, eps, layer_norm, scale_tab, shift_tab, scale_emb, shift_emb, emb_stride,
                           (hipStream_t)stream, (unsigned char*)codes, ldq, scale);
}

int ltx2_qknorm_rope(void* buf, int64_t ld, int rows, int D, int head_dim, int q_off, const float* q_weight,
                     int k_off, const float* k_weight, float eps, const float* cos, const float* sin, void* stream) {
    LTX2_CHECK_ARG(buf && q_weight, "qknorm_rope: null operand");
    const int offs[2] = {q_off, k_off};
    const float* wts[2] = {q_weight, k_weight};
    return qknorm_rope_launch((bf16*)buf, ld, rows, D, head_dim, k_weight ? 2 : 1, offs, wts, eps, cos, sin,
                              (hipStream_t)stream);
}

int ltx2_vt_transpose(const void* V, int64_t ld, void* VT, int Nkv, int Npad, int H, int head_dim, void* stream) {
    LTX2_CHECK_ARG(V && VT, "vt_transpose: null operand");
    return vt_transpose_launch((const bf16*)V, ld, (bf16*)VT, Nkv, Npad, H, (hipStream_t)stream, head_dim);
}

int ltx2_attn_head_gate(void* att, int64_t ld, const void* x, int64_t ldx, const void* gate_w, const float* gate_b,
                        float* logits, int rows, int Dq, int H, int head_dim, void* stream) {
    LTX2_CHECK_ARG(att && x && gate_w && gate_b && logits, "attn_head_gate: null operand");
    int rc = gate_logits_launch((const bf16*)x, ldx, (const bf16*)gate_w, gate_b, logits, H, rows, Dq, H, (hipStream_t)stream);
    if (rc != LTX2_OK) return rc;
    return head_gate_launch((bf16*)att, ld, logits, H, rows, H, head_dim, (hipStream_t)stream);
}

int ltx2_flash_attn_gated(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                          int H, int head_dim, float scale, const float* gate_logits, int gate_ld, void* stream) {
    LTX2_CHECK_ARG(Q && K && VT && out && gate_logits && gate_ld >= H, "flash_attn_gated: null operand / gate_ld < H");
    LTX2_CHECK_ARG(head_dim == 128 || head_dim == 64, "flash_attn_gated: head_dim=%d, only 128 and 64 are implemented", head_dim);
    AttnParams a{};
    a.Q = (const bf16*)Q;
    a.ldq = ldq;
    a.K = (const bf16*)K;
    a.ldk = ldk;
    a.VT = (const bf16*)VT;
    a.vt_head_stride = (long)head_dim * Npad;
    a.head_dim = head_dim;
    a.O = (bf16*)out;
    a.ldo = ldo;
    a.Nq = Nq;
    a.Nkv = Nkv;
    a.Npad = Npad;
    a.H = H;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.gate = gate_logits;
    a.gate_ld = gate_ld;
    return attn_launch(a, (hipStream_t)stream);
}

int ltx2_flash_attn_gated_parts(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                                int H, int head_dim, float scale, const void* x, int64_t ldx, const void* gate_w, const float* gate_b, int Dq, float* parts,
                                void* stream) {
    LTX2_CHECK_ARG(Q && K && VT && out && x && gate_w && gate_b && parts, "flash_attn_gated_parts: null operand");
    LTX2_CHECK_ARG(head_dim == 128 || head_dim == 64, "flash_attn_gated_parts: head_dim=%d, only 128 and 64 are implemented", head_dim);
    if (const int rc = gate_logits_parts_launch((const bf16*)x, ldx, (const bf16*)gate_w, parts, Nq, Dq, H, (hipStream_t)stream)) return rc;
    AttnParams a{};
    a.Q = (const bf16*)Q;
    a.ldq = ldq;
    a.K = (const bf16*)K;
    a.ldk = ldk;
    a.VT = (const bf16*)VT;
    a.vt_head_stride = (long)head_dim * Npad;
    a.head_dim = head_dim;
    a.O = (bf16*)out;
    a.ldo = ldo;
    a.Nq = Nq;
    a.Nkv = Nkv;
    a.Npad = Npad;
    a.H = H;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.gate = parts;
    a.gate_ld = H;
    a.gate_parts = GATE_LOGIT_PARTS;
    a.gate_bias = gate_b;
    return attn_launch(a, (hipStream_t)stream);
}

int ltx2_flash_attn(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out,
                    int64_t ldo, int Nq, int Nkv, int H, int head_dim, float scale, void* stream) {
    LTX2_CHECK_ARG(Q && K && VT && out, "flash_attn: null operand");
    LTX2_CHECK_ARG(head_dim == 128 || head_dim == 64, "flash_attn: head_dim=%d, only 128 and 64 are implemented", head_dim);
    AttnParams a{};
    a.Q = (const bf16*)Q;
    a.ldq = ldq;
    a.K = (const bf16*)K;
    a.ldk = ldk;
    a.VT = (const bf16*)VT;
    a.vt_head_stride = (long)head_dim * Npad;
    a.head_dim = head_dim;
    a.O = (bf16*)out;
    a.ldo = ldo;
    a.Nq = Nq;
    a.Nkv = Nkv;
    a.Npad = Npad;
    a.H = H;
    a.scale_log2e = scale * 1.4426950408889634f;
    return attn_launch(a, (hipStream_t)stream);
}

int ltx2_timestep_sinusoid(const float* t, int64_t t_stride, float mult, int T, int dim, float* out_f32,
                           void* out_bf16, void* stream) {
    LTX2_CHECK_ARG(t && (out_f32 || out_bf16), "timestep_sinusoid: null operand");
    return timestep_sinusoid_launch(t, t_stride, 0.f, mult, T, dim, out_f32, (bf16*)out_bf16, (hipStream_t)stream);
}

int ltx2_dequant_fp8_e4m3fn(const void* in, float scale, void* out_bf16, int64_t n, void* stream) {
    return dequant_fp8_launch((const unsigned char*)in, scale, (bf16*)out_bf16, n, (hipStream_t)stream);
}

int ltx2_groupnorm_silu(const void* x, const void* res, void* y, int64_t P, int C, int groups, float eps,
                        const float* gamma, const float* beta, float* scratch, int act, void* stream) {
    return groupnorm_silu_launch((const bf16*)x, (const bf16*)res, (bf16*)y, P, C, groups, eps, gamma, beta, scratch, act, (hipStream_t)stream);
}

int ltx2_s2d_downsample(const void* y, const void* x, void* out, int T, int H, int W, int Cc, int Cin, int st, int sh,
                        int sw, void* stream) {
    return s2d_downsample_launch((const bf16*)y, (const bf16*)x, (bf16*)out, T, H, W, Cc, Cin, st, sh, sw, (hipStream_t)stream);
}

int ltx2_latent_normalize_nchw(const void* x, const float* mean, const float* std, float* out, int C, int64_t P, void* stream) {
    return latent_normalize_nchw_launch((const bf16*)x, mean, std, out, C, P, (hipStream_t)stream);
}

int ltx2_rope_tables(const float* positions, const float* freq_grid, const float* max_pos, int N, int n_dims, int n_freq,
                     int half_dim, float* cos_out, float* sin_out, void* stream) {
    return rope_tables_launch(positions, freq_grid, max_pos, N, n_dims, n_freq, half_dim, cos_out, sin_out, (hipStream_t)stream);
}

int ltx2_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
    LTX2_CHECK_ARG(in && out, "cast: null operand");
    return cast_f32_bf16_launch(in, (bf16*)out, n, (hipStream_t)stream);
}

int ltx2_x0_from_velocity(const float* latent, const float* velocity, const float* ts_ptr, int64_t ts_stride,
                          float ts_scalar, float* x0, int rows, int C, void* stream) {
    LTX2_CHECK_ARG(latent && velocity && x0, "x0_from_velocity: null operand");
    return x0_from_velocity_launch(latent, velocity, ts_ptr, ts_stride, ts_scalar, x0, rows, C, (hipStream_t)stream);
}

int ltx2_euler_step(const float* x, const float* x0, const float* mask, const float* clean, float sigma,
                    float sigma_next, float* out, int rows, int C, void* stream) {
    LTX2_CHECK_ARG(x && x0 && out, "euler_step: null operand");
    return euler_step_launch(x, x0, mask, clean, sigma, sigma_next, out, rows, C, (hipStream_t)stream);
}

int ltx2_vae_prepare_latent(const float* latent, const float* std, const float* mean, const float* noise,
                            float noise_scale, void* out_bf16, int C, int64_t P, void* stream) {
    LTX2_CHECK_ARG(latent && std && mean && out_bf16, "vae_prepare_latent: null operand");
    return vae_prepare_latent_launch(latent, std, mean, noise, noise_scale, (bf16*)out_bf16, C, P, (hipStream_t)stream);
}

int ltx2_pixnorm_mod_silu(const void* x, void* y, int64_t P, int C, float eps, const float* table, const float* te,
                          int shift_row, int scale_row, void* stream) {
    LTX2_CHECK_ARG(x && y && table, "pixnorm_mod_silu: null operand");
    return pixnorm_mod_silu_launch((const bf16*)x, (bf16*)y, P, C, eps, table, te, shift_row, scale_row,
                                   (hipStream_t)stream);
}

int ltx2_vae_unpatchify(const void* x, float* video, int T, int H, int W, void* stream) {
    LTX2_CHECK_ARG(x && video, "vae_unpatchify: null operand");
    return vae_unpatchify_launch((const bf16*)x, video, T, H, W, (hipStream_t)stream);
}

int ltx2_video_chunk_to_uint8(const float* cur, const float* prev, const float* ramp, uint8_t* frames, int Tc, int prev_T, int ov, int H,
                              int W, int t_dst0, int T_out, void* stream) {
    LTX2_CHECK_ARG(cur && frames, "video_chunk_to_uint8: null operand");
    return video_chunk_to_uint8_launch(cur, prev, ramp, frames, Tc, prev_T, ov, H, W, t_dst0, T_out, (hipStream_t)stream);
}

int ltx2_tile_blend_accumulate(const float* tile, int dt, int dh, int dw, int nt, int nh, int nw, const float* mt, const float* mh,
                               const float* mw, float* out, float* wsum, int OT, int OH, int OW, int t0, int h0, int w0, void* stream) {
    LTX2_CHECK_ARG(tile && mt && mh && mw && out && wsum, "tile_blend_accumulate: null operand");
    return tile_blend_accumulate_launch(tile, dt, dh, dw, nt, nh, nw, mt, mh, mw, out, wsum, OT, OH, OW, t0, h0, w0, (hipStream_t)stream);
}

int ltx2_tile_blend_finish(float* out, const float* wsum, int64_t plane, void* stream) {
    LTX2_CHECK_ARG(out && wsum && plane > 0, "tile_blend_finish: bad argument");
    return tile_blend_finish_launch(out, wsum, (long)plane, (hipStream_t)stream);
}

int ltx2_video_to_uint8(const float* video, uint8_t* frames, int T, int H, int W, void* stream) {
    LTX2_CHECK_ARG(video && frames, "video_to_uint8: null operand");
    return video_to_uint8_launch(video, frames, T, H, W, (hipStream_t)stream);
}

}  // extern "C"
