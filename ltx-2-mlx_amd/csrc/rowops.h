// HBM-bound row / elementwise kernels of the DiT step and the VAE decoder: host launchers.
#pragma once
#include "common.h"

// out_bf16[row][:] = norm(x_f32[row][:]) * (1 + scale) + shift
//   norm  = RMS (layer_norm == 0) or mean-centred LayerNorm without affine (layer_norm == 1)
//   scale = scale_tab[d] + scale_emb[row*emb_stride + d]   (either pointer may be null -> 0)
//   shift = shift_tab[d] + shift_emb[row*emb_stride + d]
// all four null -> plain normalisation.
int norm_mod_launch(const float* x, long ldx, bf16* out, long ldo, int rows, int D, float eps, int layer_norm,
                    const float* scale_tab, const float* shift_tab, const float* scale_emb, const float* shift_emb,
                    long emb_stride, hipStream_t stream, unsigned char* q8 = nullptr, long ldq = 0, float* qscale = nullptr);
// q8 / qscale (fp8 compute path): the row additionally (out may then be null: instead) leaves as per-token e4m3fn codes + scale,
// quantised from the bf16-ROUNDED outputs exactly as quant_rows_fp8_launch would quantise `out`

// out_g[row][:] = rms_norm(x[row][:]) * (1 + scale_tab[g] + scale_emb[g]) + shift_tab[g] + shift_emb[g] for g = 0, 1 (row-invariant tables; any
// pointer may be null): two modulations of one normalised stream from ONE read of x (the AudioVideo block's cross-modal attention inputs)
// comb[l][i] = tab[l][i] + emb[i] for l < layers, i < n (round 4: every layer's AdaLN rows for one step in one launch)
int adaln_combine_launch(const float* tab, const float* emb, float* out, int layers, long n, hipStream_t stream);
int norm_mod2_launch(const float* x, long ldx, bf16* out0, bf16* out1, long ldo, int rows, int D, float eps, const float* const* scale_tab,
                     const float* const* shift_tab, const float* const* scale_emb, const float* const* shift_emb, hipStream_t stream);

// In-place on bf16 rows: for each of nseg segments (q, k) at column offsets seg_off[i] of width D:
//   y = x * rsqrt(mean(x^2) + eps) * weight_i ;  then (if cos != null) SPLIT RoPE per head:
//   pairs (h*hd + j, h*hd + hd/2 + j) rotated with cos/sin[row][h*hd/2 + j].
int qknorm_rope_launch(bf16* buf, long ld, int rows, int D, int head_dim, int nseg, const int* seg_off,
                       const float* const* weights, float eps, const float* cos, const float* sin, hipStream_t stream);

// out_bf16[row][d] = ctx_bf16[row][d] * (1 + scale_tab[d] + scale_emb[d]) + shift_tab[d] + shift_emb[d]
// (V2.3 prompt modulation of the text context, transformer.py:441-451)
int ctx_mod_launch(const bf16* ctx, bf16* out, int rows, int D, const float* scale_tab, const float* shift_tab,
                   const float* scale_emb, const float* shift_emb, hipStream_t stream);

// logits_f32[M][H] = X_bf16[M][K] @ Wg_bf16[H][K]^T + bg   (to_gate_logits, H <= 32 heads)
int gate_logits_launch(const bf16* X, long ldx, const bf16* Wg, const float* bg, float* out, long ldo, int M, int K, int H,
                       hipStream_t stream);
// the same for many rows, as GATE_LOGIT_PARTS partial sums over K slices: parts[ks][M][H] fp32, NO bias; the attention kernel's epilogue adds the parts and
// the bias in a fixed order (AttnParams::gate_parts)
constexpr int GATE_LOGIT_PARTS = 8;
int gate_logits_parts_launch(const bf16* X, long ldx, const bf16* Wg, float* parts, int M, int K, int H, hipStream_t stream);

// att[row][h*hd + j] *= 2 * sigmoid(logits[row*ldl + h])   (per-head attention gates, attention.py:241-249)
int head_gate_launch(bf16* att, long ld, const float* logits, long ldl, int rows, int H, int hd, hipStream_t stream);

// SPLIT-RoPE cos/sin tables fp32 [N][half] from positions [n_dims][N][2] (start, end), the frequency grid [n_freq]
// and max_pos [n_dims]; slot = pad + f*n_dims + d with the identity padding in front (rope.py:214-328)
int rope_tables_launch(const float* pos, const float* grid, const float* max_pos, int N, int n_dims, int n_freq, int half,
                       float* cosb, float* sinb, hipStream_t stream);

// [cos | sin] sinusoid of reference get_timestep_embedding(flip_sin_to_cos=True, shift=0), dim 256.
// t_scaled = (t ? t[i*t_stride] : t_scalar) * mult.  Writes fp32 (out_f32) and/or bf16 (out_bf16) [T][dim].
int timestep_sinusoid_launch(const float* t, long t_stride, float t_scalar, float mult, int T, int dim, float* out_f32,
                             bf16* out_bf16, hipStream_t stream);

int cast_f32_bf16_launch(const float* in, bf16* out, long n, hipStream_t stream);
// key mask fp32 [S] (non-zero = may be attended) -> 64-bit words, bit i of word t = key 64 t + i (attention.h: AttnParams::kmask)
int keymask_words_launch(const float* mask, int S, unsigned long long* words, int nwords, hipStream_t stream);
// fp8 compute path: per-row e4m3fn quantisation of bf16 rows: scale[r] = max|x[r]| / 448 (1 for a zero row),
// out[r][k] = e4m3fn_rne(x[r][k] * (1 / scale[r]))
int quant_rows_fp8_launch(const bf16* x, long ldx, int rows, int K, unsigned char* out, long ldo, float* scale, hipStream_t stream);
// out_bf16 = bf16(f32(fp8_e4m3fn) * scale): checkpoint weights quantised with a per-tensor weight_scale
int dequant_fp8_launch(const unsigned char* in, float scale, bf16* out, long n, hipStream_t stream);

// x0[row][c] = latent[row][c] - ts(row) * vel[row][c]   (ts = ts_ptr[row*ts_stride] if ts_ptr else ts_scalar)
int x0_from_velocity_launch(const float* latent, const float* vel, const float* ts_ptr, long ts_stride, float ts_scalar,
                            float* x0, int rows, int C, hipStream_t stream);

// x0' = mask ? x0*mask(row) + clean*(1-mask(row)) : x0 ;  out = x + (x - x0')/sigma * (sigma_next - sigma)
int euler_step_launch(const float* x, const float* x0, const float* mask, const float* clean, float sigma,
                      float sigma_next, float* out, int rows, int C, hipStream_t stream);

// ---- spatial upscaler (channels-last bf16 [P][C]) ----
// y = [silu]( GroupNorm_G(x over (C/G, all positions)) * gamma + beta + res );  scratch: 2*G*(1 + ceil(P/16)) floats
int groupnorm_silu_launch(const bf16* x, const bf16* res, bf16* y, long P, int C, int G, float eps, const float* gamma,
                          const float* beta, float* scratch, int act, hipStream_t stream);
// VAE encoder downsample tail: out = space_to_depth(y) + group_mean(space_to_depth(x))   (see ltx2hip.h)
int s2d_downsample_launch(const bf16* y, const bf16* x, bf16* out, int T, int H, int W, int Cc, int Cin, int st, int sh, int sw,
                          hipStream_t stream);
// out fp32 [C][P] = (x[P][C] - mean[c]) / std[c]
int latent_normalize_nchw_launch(const bf16* x, const float* mean, const float* stdv, float* out, int C, long P, hipStream_t stream);

// ---- VAE decoder elementwise ops (channels-last bf16 activations [P][C]) ----
// latent fp32 NCTHW [C][P] -> bf16 [P][C]: v = latent*std[c] + mean[c]; if noise_scale>0: v = noise*ns + (1-ns)*v
int vae_prepare_latent_launch(const float* latent, const float* std, const float* mean, const float* noise,
                              float noise_scale, bf16* out, int C, long P, hipStream_t stream);
// y = silu( x * rsqrt(mean_c(x^2)+eps) * (1 + scale) + shift ), scale/shift = tab[row_idx*C + c] (+ te[row_idx*C + c])
int pixnorm_mod_silu_launch(const bf16* x, bf16* y, long P, int C, float eps, const float* tab, const float* te,
                            int shift_row, int scale_row, hipStream_t stream);
// same, writing the PADDED volume [T+2][H+2][W+2][C] (replicate T with pad_front leading frames, reflect H / W) that the
// implicit-GEMM conv of gemm_v4.hip reads
int pixnorm_mod_silu_padded_launch(const bf16* x, bf16* y, int T, int H, int W, int C, float eps, const float* tab, const float* te,
                                   int shift_row, int scale_row, int pad_front, hipStream_t stream);
// conv_out [T][H][W][48] bf16 -> video fp32 [3][T][4H][4W]  (reference ops.unpatchify packing (c, r_w, r_h))
int pad_volume_launch(const bf16* x, bf16* y, int T, int H, int W, int C, int pad_front, hipStream_t stream);
int vae_unpatchify_launch(const bf16* x, float* video, int T, int H, int W, hipStream_t stream);
// video fp32 [3][T][H][W] -> frames uint8 [T][H][W][3] = trunc(clip((v+1)/2,0,1)*255)
int video_to_uint8_launch(const float* video, unsigned char* frames, int T, int H, int W, hipStream_t stream);
int video_chunk_to_uint8_launch(const float* cur, const float* prev, const float* ramp, unsigned char* frames, int Tc, int prev_T, int ov,
                                int H, int W, int t_dst0, int T_out, hipStream_t stream);
int tile_blend_accumulate_launch(const float* tile, int dt, int dh, int dw, int nt, int nh, int nw, const float* mt, const float* mh,
                                 const float* mw, float* out, float* wsum, int OT, int OH, int OW, int t0, int h0, int w0, hipStream_t stream);
int tile_blend_finish_launch(float* out, const float* wsum, long plane, hipStream_t stream);
