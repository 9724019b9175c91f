/*
 * ltx2hip.h -- C ABI of libltx2hip.so: the MI355X (gfx950) implementation of the LTX-2
 * denoise (DiT) + CausalVideoVAE decode hot path.
 *
 * This is the drop-in boundary.  The reference (Acelogic/LTX-2-MLX) is pure Python over
 * Apple MLX; the entry points below are what its Python layer would bind (via ctypes, see
 * INTEGRATION.md) in place of the MLX calls cited next to each function.  All paths are
 * relative to /root/reference.
 *
 * Conventions
 *   - every function returns int: 0 = LTX2_OK, negative = LTX2_E_* ; nothing throws across the
 *     boundary; ltx2_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers (HBM) unless named host_*; the library never allocates or
 *     frees caller memory and never retains input/output pointers beyond the call, except
 *     weights (owned by the caller for the lifetime of the context) and the bound workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     enqueues work on that stream; no internal threads; one context per GPU / process.
 *   - activations: bf16 = raw uint16 bfloat16; fp32 where stated.  Batch is 1 (the reference
 *     hard-wires batch = 1: pipelines/distilled.py:314, scripts/generate.py:1768).
 */
#ifndef LTX2HIP_H
#define LTX2HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTX2_OK 0
#define LTX2_E_INVALID (-1) /* bad argument / unsupported shape  (Python side raises ValueError) */
#define LTX2_E_HIP (-2)     /* HIP runtime error                   (RuntimeError) */
#define LTX2_E_STATE (-3)   /* call order, missing weight, small workspace (RuntimeError) */

#define LTX2_DTYPE_BF16 0
#define LTX2_DTYPE_F32 1
#define LTX2_DTYPE_FP8_E4M3FN 2   /* ltx2_dit_set_weight: codes of a fp8-resident linear weight; `<name>_scale` (fp32 [out]) must be set too */

/* Version of THIS header's signatures.  ltx2_abi_version() returns the version the library was built with; a caller compares the
 * two before its first compute call (the Python binding refuses to load a library that reports another version).  History:
 * 1 = round 1;  2 = round 2 added the `sigma` / `sigma_dev` argument to ltx2_dit_forward / ltx2_dit_denoise_step (in the middle of
 * the list: an old caller's arguments would shift silently), round 3 added ltx2_dit_health; round 4 added entry points only
 * (ltx2_clear_error, ltx2_adaln_rmsnorm2, ltx2_flash_attn_gated, ltx2_flash_attn_form, ltx2_dit_graph_capture_cond[_av], the
 * "adaln_combine" option) without bumping the version -- a round-3 library then passed the check and failed later on a symbol lookup;
 * 3 = round 5: the round-4 additions are part of the version, ltx2_flash_attn_form (the 64-rows-per-wave experiment's entry), ltx2_flash_attn_ws /
 * ltx2_flash_attn_workspace_bytes (the stream-K launch form) and ltx2_dit_health (its timeout flag) are REMOVED with their kernels, and
 * ltx2_dit_graph_capture_cond[_av] take the element counts of the mask / clean-latent buffers they replay from.  Rule from here on: any change
 * of the exported symbol set or of a signature bumps the version. */
#define LTX2_ABI_VERSION 3

const char* ltx2_last_error(void);
/* forget the calling thread's message (a binding that loads several builds of the library reads every build's message after a failure and
 * clears them all, so a stale message of one build is never reported for a failure in another; round 4, additive) */
void ltx2_clear_error(void);
int ltx2_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Per-kernel entry points (unit parity against the oracle)
 * ------------------------------------------------------------------------------------------ */

/* GEMM epilogues */
#define LTX2_EPI_BF16 0            /* out_bf16 = acc + bias */
#define LTX2_EPI_GELU_BF16 1       /* out_bf16 = gelu_tanh(acc + bias) */
#define LTX2_EPI_SILU_BF16 2       /* out_bf16 = silu(acc + bias) */
#define LTX2_EPI_F32 3             /* out_f32  = acc + bias */
#define LTX2_EPI_RESID_GATE_F32 4  /* out_f32 += gate * (acc + bias) */
#define LTX2_EPI_ADD_BF16 5        /* out_bf16 = acc + bias + res_bf16 */

/* out[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias).  Replaces mlx nn.Linear on the DiT path:
 * model/transformer/attention.py:225-228,253 (to_q/k/v/out), feed_forward.py:23,49 (FFN),
 * model.py:49-56 (caption projection), model.py:242 (patchify_proj), model.py:757 (proj_out),
 * and the fused helpers transformer.py:35-46 (_compiled_residual_gate: epilogue 4) and
 * feed_forward.py:26 (gelu_approx: epilogue 1).
 * gate (epilogue 4): gate[m*gate_stride + n] (+ gate_table[n]); both NULL -> 1.            */
int ltx2_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M,
                   int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table,
                   const void* res, int64_t ldres, void* stream);

/* Fused QKV projection (attention.py:196-214: to_q / to_k / to_v on the same input): out[M][N] = A @ W^T + bias for the columns
 * < vt_col0 (Q and K), while the columns >= vt_col0 (V, heads of head_dim) leave as attention's V^T operand vt[H][head_dim][Npad]
 * with the key order ltx2_vt_transpose produces (rows >= M zero up to Npad) -- written by the GEMM's epilogue on the shapes the
 * 4-wave kernel takes (M >= 1024, N % 256 == 0, vt_col0 % 256 == 0, K % 128 == 0; *fused = 1), else by a transpose pass after
 * the GEMM (*fused = 0; the V columns of `out` are then written too).  Bit-identical to ltx2_gemm_bf16 + ltx2_vt_transpose. */
int ltx2_gemm_qkv_vt(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                     void* vt, int vt_col0, int Npad, int head_dim, int* fused, void* stream);

/* fp8-RESIDENT weights (reference loader/fp8_loader.py:14-51,54-130 dequantises at load; BASELINE config 3): W8 = float8_e4m3fn
 * codes [N][K], wscale[N] fp32 = the checkpoint's per-tensor `weight_scale` repeated per output row (fused q/k/v carry one
 * value per part).  The kernel expands bf16(f32(code) * wscale[n]) on the way from LDS to the MFMA -- exactly
 * ltx2_dequant_fp8_e4m3fn's arithmetic -- so out is BIT-IDENTICAL to ltx2_gemm_bf16 on the dequantised weights.
 * N % 256 == 0, K % 128 == 0, K >= 256; epilogues BF16 / GELU_BF16 / F32 / RESID_GATE_F32.                          */
int ltx2_gemm_w8a16(const void* A, int64_t lda, const void* W8, const float* wscale, const float* bias, void* out, int64_t ldo, int M,
                    int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table, void* stream);

/* Text cross-attention with q_norm folded in (attention.py:231-237 applied as arithmetic instead of a pass over q; round 3).
 * ltx2_gemm_bf16_rowss: out = A @ W^T + bias (bf16) and, where the 4-wave kernel takes the shape (*written = 1), rowss[m][N / 64] = the
 * sums of squares of out's ROUNDED values over each 64-column strip of row m; *written = 0: plain GEMM, rowss untouched.
 * ltx2_flash_attn_rowscale: ltx2_flash_attn with a per-query-row scale: row q uses scale * rsqrt(sum_j q_ss[q][j] / q_norm_dim + q_eps),
 * i.e. softmax((rms_norm(Q) K'^T) * scale) V for K' carrying q_norm.weight * k_norm.weight -- identical to normalising Q first up to the
 * rounding of the normalised Q to 16 bits (which this form does not do).  head_dim 128, q_ss_ld % 16 == 0 (ABI 3: the row's four lanes read f32x4 quarters; rows 64-byte aligned).                     */
int ltx2_gemm_bf16_rowss(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K, float* rowss,
                         int* written, void* stream);
int ltx2_flash_attn_rowscale(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                             int H, int head_dim, float scale, const float* q_ss, int q_ss_ld, int q_norm_dim, float q_eps, void* stream);

/* An RMS norm folded around two GEMMs (round 6; rms_norm + nn.Linear, transformer.py:217-226, as arithmetic instead of a pass over x):
 * rms_norm(x) (1 + s) in front of a projection (W, b) equals r (x (1 + s)) W^T + b, r[m] = rsqrt(mean_j x[m][j]^2 + eps).
 * ltx2_gemm_bf16_fold: ltx2_gemm_bf16 with the producer / consumer half of that identity:
 *   epilogue RESID_GATE_F32 (x += gate_table * (acc + bias), row-invariant gate): shadow[m][n] = 16-bit(x_new[m][n] * (1 + shadow_scale[n]))
 *     (row stride ld_shadow; shadow_scale null: * 1) and shadow_ss[(n / 256) * ld_ss + m] = the sum of x_new[m][n]^2 over each 256-column tile
 *     (ld_ss >= the row tiles' extent: M rounded up to a multiple of 256 is always enough; % 4 == 0);
 *   epilogue BF16 / GELU_BF16: out = epilogue(r[m] * acc + bias) with r[m] = rsqrt(sum_{j < rf_nparts} rf_parts[j * rf_ld + m] / rf_dim + rf_eps)
 *     formed inside the kernel (rf_parts = a producer's shadow_ss, rf_nparts <= 24).
 *   *supported = 0 (nothing launched) when the 4-wave layout-3 kernel does not take the problem (M >= 1024, N % 256 == 0, K % 128 == 0, dense 16-bit
 *   weights) or neither half is asked for.                                                                                                         */
int ltx2_gemm_bf16_fold(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                        const float* gate_table, void* shadow, int64_t ld_shadow, const float* shadow_scale, float* shadow_ss, int64_t ld_ss,
                        const float* rf_parts, int64_t rf_ld, int rf_nparts, int rf_dim, float rf_eps, int* supported, void* stream);

/* flash attention with a key mask (attention.py:38-70 with the additive mask model.py:163-201 builds from a boolean (B, S) context
 * mask): mask fp32 [Nkv], non-zero = the key may be attended; a masked key takes no weight unless every key is masked (then the
 * row averages V over all keys, as the reference's -finfo.max bias does).  words: scratch of Npad / 8 bytes on the device.        */
int ltx2_flash_attn_keymask(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo,
                            int Nq, int Nkv, int H, int head_dim, float scale, const float* mask, void* words, void* stream);

/* Which kernel ltx2_gemm_bf16 / ltx2_gemm_w8a16 / ltx2_gemm_fp8 (weights = 0 / 1 / 2) would run for a dense M x N x K problem with
 * this epilogue -- host logic only, nothing is launched, no GPU needed.  Returns LTX2_ROUTE_* (negative: unsupported), OR-ed with
 * 0x100 when has_vt != 0 and the fused-QKV V^T output (ltx2_gemm_qkv_vt with N = 3 * inner_dim) would come from the GEMM's own epilogue.
 * The dispatch is a parity surface: tests pin the route of every GEMM the models issue.                                  */
#define LTX2_ROUTE_SKINNY 0     /* gemm_skinny.hip: M <= 128 rows, weight streaming                         */
#define LTX2_ROUTE_V4_224 1     /* gemm_v4.hip, generated asm K loop, 224 x 256 tiles                       */
#define LTX2_ROUTE_V4_256 2     /*                                    256 x 256 tiles                       */
#define LTX2_ROUTE_V4_W8_224 3  /* ... fp8-resident weights expanded to bf16 in the loop                     */
#define LTX2_ROUTE_V4_W8_256 4
#define LTX2_ROUTE_V4_F8_224 5  /* ... fp8 compute (v_mfma_f32_32x32x64_f8f6f4)                             */
#define LTX2_ROUTE_V4_F8_256 6
#define LTX2_ROUTE_PP 7         /* gemm_pp.hip: 256 x 256 ping-pong                                         */
#define LTX2_ROUTE_SMALL 8      /* gemm.hip: 128 x 128                                                      */
#define LTX2_ROUTE_NARROW 9     /* gemm.hip: 128 x 64 (N <= 64)                                             */
int ltx2_gemm_route(int M, int N, int K, int epilogue, int weights, int has_vt);

/* fp8 COMPUTE (BASELINE config 3, "fp8 weights (CDNA4 fp8 MFMA)"; opt-in, not the parity-exact default): both operands are
 * float8_e4m3fn codes, the products run on v_mfma_f32_32x32x64_f8f6f4 at twice the bf16 MFMA rate, fp32 accumulation;
 *   out = epilogue(ascale[m] * wscale[n] * sum_k f32(A8[m][k]) * f32(W8[n][k]) + bias[n]).
 * ltx2_quantize_rows_fp8 is the quantiser of both sides: scale[r] = max_k |x[r][k]| / 448 (1 for an all-zero row), codes[r][k] =
 * e4m3fn_rne(x[r][k] * (1 / scale[r])) -- per TOKEN for activations, per OUTPUT CHANNEL for weights quantised at load (an fp8
 * checkpoint's codes + per-tensor `weight_scale`, reference loader/fp8_loader.py:14-51, are used as they are).
 * N % 256 == 0, K % 256 == 0, K >= 512, lda % 16 == 0; epilogues BF16 / GELU_BF16 / F32 / RESID_GATE_F32.                    */
int ltx2_quantize_rows_fp8(const void* x_bf16, int64_t ldx, int rows, int K, void* codes, int64_t ldo, float* scale, void* stream);
int ltx2_gemm_fp8(const void* A8, int64_t lda, const float* ascale, const void* W8, const float* wscale, const float* bias, void* out,
                  int64_t ldo, int M, int N, int K, int epilogue, const float* gate, int64_t gate_stride, const float* gate_table,
                  void* stream);

/* ltx2_gemm_qkv_vt on the fp8 compute path (same contract: V columns leave as attention's V^T operand). */
int ltx2_gemm_fp8_qkv_vt(const void* A8, int64_t lda, const float* ascale, const void* W8, const float* wscale, const float* bias, void* out,
                         int64_t ldo, int M, int N, int K, void* vt, int vt_col0, int Npad, int head_dim, int* fused, void* stream);

/* Skinny fp32-activation path (M <= 16): out_f32 = act_out(act_in(a) @ W^T + bias); act: 0 none,
 * 1 silu, 2 gelu_tanh.  Replaces TimestepEmbedding / AdaLayerNormSingle linears for a scalar
 * sigma (model/transformer/timestep_embedding.py:112-124,187-202) and the VAE TimestepEmbedder
 * (model/video_vae/simple_decoder.py:54-59).                                                 */
int ltx2_gemv_f32(const float* a, int64_t lda, const void* W, const float* bias, float* out, int64_t ldo, int M,
                  int N, int K, int act_in, int act_out, void* stream);

/* 3x3x3 stride-1 conv3d on channels-last bf16 activations x[T][H][W][Cin], weights
 * w[Cout][27][Cin] (tap = (kt*3+kh)*3+kw), reflect pad H/W, replicate pad T (causal: 2 front).
 * Replaces Conv3dSimple.__call__ (model/video_vae/simple_decoder.py:90-180).
 *   mode 0: out[T][H][W][Cout] = conv + bias
 *   mode 1: out = conv + bias + res          (ResBlock3d skip, simple_decoder.py:240)
 *   mode 2: depth-to-space upsample epilogue (DepthToSpaceUpsample3d, simple_decoder.py:287-313):
 *           weight rows must be pre-permuted to n' = s*Cf + c (s = (a*fh+b)*fw+d, Cf = Cout/(ft*fh*fw));
 *           out[T*ft - (ft>1)][H*fh][W*fw][Cf]; residual != 0 adds tile(d2s(x)).                */
int ltx2_conv3d_fused(const void* x, const void* w, const float* bias, void* out, int T, int H, int W, int Cin,
                      int Cout, int causal, int mode, const void* res, int ft, int fh, int fw, int residual,
                      int pad_zero, int kt, void* stream);
/* pad_zero = 1: zero padding in T/H/W instead of reflect/replicate (spatial upscaler, upscaler/spatial.py:20-87);
 * pad_zero = 2: zero padding in H/W, replicated temporal edge (VAE encoder Conv3dSimple, simple_encoder.py:44-75);
 * kt = 1: per-frame 3x3 conv2d (weight [Cout][9*Cin]); with mode 2, ft=1, fh=fw=2 the epilogue is
 * PixelShuffle(2) (SpatialRationalResampler, upscaler/spatial.py:267-323).                               */

/* GroupNorm over (C/groups, T, H, W) on channels-last bf16 x[P][C] (upscaler/spatial.py:89-128), fused with the
 * affine, an optional residual add and SiLU:  y = silu(gn(x) * gamma + beta + res).  scratch: 2*groups*(1 +
 * ceil(P/16)) floats (two-level reduction without atomics: bit-reproducible).  act = 0 skips the SiLU.   */
int ltx2_groupnorm_silu(const void* x, const void* res, void* y, int64_t P, int C, int groups, float eps,
                        const float* gamma, const float* beta, float* scratch, int act, void* stream);

/* VAE encoder SpaceToDepthDownsample3d tail (simple_encoder.py:183-257): y = conv(x') [T][H][W][Cc] and the conv
 * input x' [T][H][W][Cin] (first frame already duplicated when st = 2) ->
 * out[T/st][H/sh][W/sw][Cc*sp] = space_to_depth(y) + group_mean(space_to_depth(x')), sp = st*sh*sw,
 * channel c*sp + (a*sh + b)*sw + d, groups of Cin*sp / (Cc*sp) consecutive space-to-depth channels.      */
int ltx2_s2d_downsample(const void* y, const void* x, void* out, int T, int H, int W, int Cc, int Cin, int st, int sh,
                        int sw, void* stream);

/* Upscaler output: x bf16 [P][C] -> out fp32 [C][P] = (x - mean[c]) / std[c]  (PerChannelStatistics.normalize,
 * video_vae/ops.py:173-186).                                                                             */
int ltx2_latent_normalize_nchw(const void* x, const float* mean, const float* std, float* out, int C, int64_t P,
                               void* stream);

/* out_bf16 = norm(x_f32) * (1 + scale) + shift ; scale = scale_tab[d] + scale_emb[row*emb_stride+d].
 * layer_norm = 0: RMS  (transformer.py:16-31 _compiled_adaln_forward; attention.py:88-100 rms_norm)
 * layer_norm = 1: LayerNorm without affine (model.py:553,744-758).  NULL pointers contribute 0.   */
int ltx2_adaln_rmsnorm(const float* x, int64_t ldx, void* out, int64_t ldo, int rows, int D, float eps,
                       int layer_norm, const float* scale_tab, const float* shift_tab, const float* scale_emb,
                       const float* shift_emb, int64_t emb_stride, void* stream);
/* Two modulations of ONE RMS-normalised stream from one read of x (round 4; the cross-modal attention's a2v / v2a inputs, transformer.py:556-620):
 * out_g = rms_norm(x) * (1 + scale_g) + shift_g, g = 0, 1 (row-invariant fp32 vectors [D], any may be null).  D <= 4096.            */
int ltx2_adaln_rmsnorm2(const float* x, int64_t ldx, void* out0, void* out1, int64_t ldo, int rows, int D, float eps, const float* scale0,
                        const float* shift0, const float* scale1, const float* shift1, void* stream);

/* The same with the fp8 compute path's per-token quantiser fused in: codes[rows][ldq] + scale[rows] = ltx2_quantize_rows_fp8 of the
 * bf16-rounded outputs, bit for bit; out_bf16 may be NULL (the GEMM that follows reads only the codes).                    */
int ltx2_adaln_rmsnorm_fp8(const float* x, int64_t ldx, void* out_bf16, int64_t ldo, void* codes, int64_t ldq, float* scale, int rows, int D,
                           float eps, int layer_norm, const float* scale_tab, const float* shift_tab, const float* scale_emb,
                           const float* shift_emb, int64_t emb_stride, void* stream);

/* In place on bf16 rows: RMSNorm(weight) over the full inner dim of q (and k), then SPLIT RoPE
 * (cos/sin fp32 [rows][D/2], slot h*hd/2 + j) if cos != NULL.
 * Replaces attention.py:231-237 + rope.py:92-144.  k_* may be NULL/absent (nseg = 1).        */
int ltx2_qknorm_rope(void* buf, int64_t ld, int rows, int D, int head_dim, int q_off, const float* q_weight,
                     int k_off, const float* k_weight, float eps, const float* cos, const float* sin, void* stream);

/* V[Nkv][ld] (head h at columns h*hd) -> VT[H][hd][Npad], keys permuted inside each block of 32
 * to match the MFMA accumulator layout of the attention kernel (ABI 3: position 8g + 4h + r of a block holds key 16h + 4g + r -- the
 * 16x16x32 kernel's order; a V^T written by an ABI-2 library is NOT valid for this one); padded keys are zero.
 * head_dim hd = 128 (video streams) or 64 (audio streams, audio<->video attention).          */
int ltx2_vt_transpose(const void* V, int64_t ld, void* VT, int Nkv, int Npad, int H, int head_dim, void* stream);

/* out[q][h*hd..] = softmax(Q_h K_h^T * scale) V_h, non-causal, no mask, head_dim 128 or 64; ldq, ldk, ldo multiples of 8 (16-byte rows).
 * Replaces _compiled_attention_core_no_mask (attention.py:12-34).                             */
int ltx2_flash_attn(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out,
                    int64_t ldo, int Nq, int Nkv, int H, int head_dim, float scale, void* stream);


/* Per-head attention gates (V2.3, attention.py:241-249): logits[rows][H] (fp32 scratch, also returned)
 * = x[rows][Dq] @ gate_w[H][Dq]^T + gate_b ; att[:, h*hd:(h+1)*hd] *= 2*sigmoid(logits[:, h]).  H <= 32. */
int ltx2_attn_head_gate(void* att, int64_t ld, const void* x, int64_t ldx, const void* gate_w, const float* gate_b,
                        float* logits, int rows, int Dq, int H, int head_dim, void* stream);

/* ltx2_flash_attn with the per-head gates applied in the kernel's epilogue (round 4: the engine's form -- the gate multiplies the fp32 result
 * before it is rounded, instead of a pass over the rounded output): out[q, h*hd:(h+1)*hd] = 2*sigmoid(gate_logits[q*gate_ld + h]) * attention. */
int ltx2_flash_attn_gated(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                          int H, int head_dim, float scale, const float* gate_logits, int gate_ld, void* stream);

/* The engine's form of the gated attention for many rows (round 5): the gate logits x[Nq][Dq] @ gate_w[H][Dq]^T leave as 8 partial sums over K slices
 * (parts: fp32 scratch [8][Nq][H]; H <= 32, Dq % 256 == 0, ldx % 8 == 0), and the attention kernel's epilogue adds the 8 values and gate_b in slice order before
 * 2 * sigmoid(.) multiplies the fp32 result -- ltx2_attn_head_gate's logits up to the summation order.                                              */
int ltx2_flash_attn_gated_parts(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* VT, int Npad, void* out, int64_t ldo, int Nq, int Nkv,
                                int H, int head_dim, float scale, const void* x, int64_t ldx, const void* gate_w, const float* gate_b, int Dq, float* parts,
                                void* stream);

/* SPLIT-RoPE tables (precompute_freqs_cis with use_middle_indices_grid, rope.py:214-328,365-418): positions fp32
 * [n_dims][N][2] = [start, end) per axis, freq_grid [n_freq] = theta^linspace(0,1,n_freq)*pi/2 (host-computed, exact),
 * max_pos [n_dims]; writes cos/sin fp32 [N][half_dim], slot = pad + f*n_dims + d, pad = half_dim - n_dims*n_freq
 * identity slots in FRONT.  Replaces the per-step table rebuild of model.py:203-229.                    */
int ltx2_rope_tables(const float* positions, const float* freq_grid, const float* max_pos, int N, int n_dims, int n_freq,
                     int half_dim, float* cos_out, float* sin_out, void* stream);

/* [cos | sin] sinusoid, dim 256 (timestep_embedding.py:10-60 with flip_sin_to_cos, shift 0;
 * simple_decoder.py:12-39).  Either output may be NULL.                                       */
int ltx2_timestep_sinusoid(const float* t, int64_t t_stride, float mult, int T, int dim, float* out_f32,
                           void* out_bf16, void* stream);

/* Load-time dequantisation of fp8 checkpoints: out_bf16[i] = bf16(f32(e4m3fn in[i]) * weight_scale)
 * (loader/weight_converter.py:391-395, loader/fp8_loader.py:14-51).                            */
int ltx2_dequant_fp8_e4m3fn(const void* in, float scale, void* out_bf16, int64_t n, void* stream);
int ltx2_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);

/* x0 = latent - ts * velocity  (X0Model.__call__, model.py:912-918); ts_ptr NULL -> ts_scalar. */
int ltx2_x0_from_velocity(const float* latent, const float* velocity, const float* ts_ptr, int64_t ts_stride,
                          float ts_scalar, float* x0, int rows, int C, void* stream);

/* post_process_latent + EulerDiffusionStep.step fused (pipelines/common.py:169-190,
 * components/diffusion_steps.py:36-67, scripts/generate.py:905-930).  sigma == 0 -> LTX2_E_INVALID
 * with message "Sigma can't be 0.0" (core_utils.py:54-55).  mask/clean may both be NULL.      */
int ltx2_euler_step(const float* x, const float* x0, const float* mask, const float* clean, float sigma,
                    float sigma_next, float* out, int rows, int C, void* stream);

/* VAE elementwise glue (simple_decoder.py:492-498, 228-238/339-342, ops.py:109-125, :792-798)  */
int ltx2_vae_prepare_latent(const float* latent, const float* std, const float* mean, const float* noise,
                            float noise_scale, void* out_bf16, int C, int64_t P, void* stream);
int ltx2_pixnorm_mod_silu(const void* x, void* y, int64_t P, int C, float eps, const float* table, const float* te,
                          int shift_row, int scale_row, void* stream);
int ltx2_vae_unpatchify(const void* x, float* video, int T, int H, int W, void* stream);
int ltx2_video_to_uint8(const float* video, uint8_t* frames, int T, int H, int W, void* stream);
/* decode_latent's chunk cross-fade, trim and uint8 conversion in one pass (simple_decoder.py:760-798): frame j of the chunk
 * `cur` [3][Tc][H][W] lands on output frame t_dst0 + j of `frames` [T_out][H][W][3]; with `prev` (the previous chunk,
 * [3][prev_T][H][W]) the first `ov` frames are prev_tail * (1 - ramp[j]) + cur * ramp[j] (ramp = torch.linspace(0, 1, ov) on the
 * device), rounded like the separate reference ops.                                                                  */
int ltx2_video_chunk_to_uint8(const float* cur, const float* prev, const float* ramp, uint8_t* frames, int Tc, int prev_T, int ov, int H,
                              int W, int t_dst0, int T_out, void* stream);
/* decode_tiled's trapezoid blend (video_vae/tiling.py:380-412): out[3][OT][OH][OW] += tile[3][dt][dh][dw](cropped to nt,nh,nw) *
 * mt[t] mh[h] mw[w] and wsum += mask at offset (t0,h0,w0); ltx2_tile_blend_finish divides by clamp(wsum, 1e-8).       */
int ltx2_tile_blend_accumulate(const float* tile, int dt, int dh, int dw, int nt, int nh, int nw, const float* mt, const float* mh,
                               const float* mw, float* out, float* wsum, int OT, int OH, int OW, int t0, int h0, int w0, void* stream);
int ltx2_tile_blend_finish(float* out, const float* wsum, int64_t plane, void* stream);

/* ------------------------------------------------------------------------------------------
 * DiT engine: LTXModel forward behind one call, VideoOnly or AudioVideo, 19B-style blocks or the
 * 22B "V2.3" variant (cross_attention_adaln + apply_gated_attention).
 * Replaces X0Model(LTXModel(...)).__call__ (model/transformer/model.py:776-881,895-936),
 * BasicTransformerBlock.__call__ (transformer.py:191-238) and BasicAVTransformerBlock.__call__
 * (transformer.py:457-648) x num_layers.
 * ------------------------------------------------------------------------------------------ */
#define LTX2_MODEL_VIDEO_ONLY 0
#define LTX2_MODEL_AUDIO_VIDEO 1

typedef struct ltx2_dit ltx2_dit;

typedef struct ltx2_dit_config {
    int num_layers;        /* 48 */
    int num_heads;         /* 32 */
    int head_dim;          /* 128 */
    int in_channels;       /* 128 */
    int out_channels;      /* 128 */
    int caption_channels;  /* 3840; 0 = no caption_projection (context already inner_dim wide) */
    float norm_eps;        /* 1e-6 */
    float timestep_scale;  /* 1000 */
    /* --- fields below default to 0 for the 19B VideoOnly model --- */
    int model_type;             /* LTX2_MODEL_VIDEO_ONLY / LTX2_MODEL_AUDIO_VIDEO (model.py:436) */
    int audio_heads;            /* 32  (LTXModel.AUDIO_ATTENTION_HEADS, model.py:429); must equal num_heads */
    int audio_head_dim;         /* 64  (AUDIO_HEAD_DIM, model.py:430) */
    int audio_in_channels;      /* 128 */
    int audio_out_channels;     /* 128 */
    int cross_attention_adaln;  /* V2.3: 9 AdaLN rows, prompt_adaln_single, prompt_scale_shift_table (transformer.py:427-455) */
    int apply_gated_attention;  /* V2.3: to_gate_logits, out *= 2*sigmoid(.) per head (attention.py:241-249) */
    float av_ca_timestep_scale; /* 1 (av_ca_timestep_scale_multiplier, model.py:452) */
} ltx2_dit_config;

int ltx2_dit_create(const ltx2_dit_config* cfg, ltx2_dit** out);
void ltx2_dit_destroy(ltx2_dit* ctx);

/* Register a weight (caller keeps it alive).  Names are the checkpoint keys after stripping
 * "model.diffusion_model." (loader/weight_converter.py:277-315), with these load-time fusions:
 *   transformer_blocks.{i}.attn1.to_qkv.{weight,bias}  = cat(to_q, to_k, to_v)   [3D, D]
 *   transformer_blocks.{i}.attn2.to_kv.{weight,bias}   = cat(to_k, to_v)         [2D, D]
 * Linear weights: bf16 [out, in]; biases, norm weights, scale_shift_tables: fp32.           */
int ltx2_dit_set_weight(ltx2_dit* ctx, const char* name, const void* ptr, int dtype, int64_t numel);

/* Workspace (activations + per-prompt caches).  per_token != 0 sizes the per-token AdaLN path.
 * AudioVideo models use the *_av forms (Na audio tokens, Sa audio-text tokens).               */
int64_t ltx2_dit_workspace_bytes(const ltx2_dit* ctx, int N, int S, int per_token);
int ltx2_dit_bind_workspace(ltx2_dit* ctx, void* ptr, int64_t bytes, int N, int S, int per_token);
int64_t ltx2_dit_workspace_bytes_av(const ltx2_dit* ctx, int N, int S, int Na, int Sa, int per_token);
int ltx2_dit_bind_workspace_av(ltx2_dit* ctx, void* ptr, int64_t bytes, int N, int S, int Na, int Sa, int per_token);

/* Per-prompt, step-invariant work hoisted out of the loop (the reference recomputes it every
 * step: model.py:262-271): caption projection (model.py:142-161), per-layer cross-attention
 * K (k_norm applied) and V^T of the projected context (attention.py:227-232), and the RoPE
 * tables cos/sin fp32 [N][D/2] computed by the caller from positions (rope.py:365-418).     */
int ltx2_dit_prepare(ltx2_dit* ctx, const float* context, int S, const float* rope_cos, const float* rope_sin,
                     void* stream);
/* AudioVideo: per modality the text context, the self-attention RoPE tables [N][D/2] and the
 * cross-modal tables [N][Da/2] (temporal axis only, audio inner dim; model.py:320-344).  With
 * cross_attention_adaln the text K/V depend on sigma and are recomputed per step instead.    */
int ltx2_dit_prepare_av(ltx2_dit* ctx, const float* v_context, int S, const float* v_cos, const float* v_sin,
                        const float* v_cross_cos, const float* v_cross_sin, const float* a_context, int Sa,
                        const float* a_cos, const float* a_sin, const float* a_cross_cos, const float* a_cross_sin,
                        void* stream);

/* velocity[N][out_channels] (fp32) = LTXModel(latent[N][in_channels] fp32, timesteps).
 * n_timesteps = 1: one sigma for all tokens (Modality.timesteps shape (B,), scripts/generate.py:1946);
 * n_timesteps = N: per-token sigma (pipelines/common.py:193-232).                           */
int ltx2_dit_forward(ltx2_dit* ctx, const float* latent, const float* timesteps, int n_timesteps, const float* sigma,
                     float* velocity, void* stream);
/* AudioVideo forward (model.py:776-881).  *_sigma: one device float per modality = Modality.sigma
 * (drives the prompt AdaLN of its own modality and the cross-modal AdaLN of the OTHER one,
 * model.py:151-161,392-404).  The VideoOnly entries take it as `sigma` / `sigma_dev` (one device float; NULL =
 * timesteps[0], which is only right when no token carries a conditioning mask: with image conditioning
 * timesteps = mask * sigma, model.py:151-158).                                                 */
int ltx2_dit_forward_av(ltx2_dit* ctx, const float* v_latent, const float* v_timesteps, int n_v_timesteps,
                        const float* v_sigma, const float* a_latent, const float* a_timesteps, int n_a_timesteps,
                        const float* a_sigma, float* v_velocity, float* a_velocity, void* stream);

/* One sampling step: forward -> x0 = latent - ts*v -> post_process -> Euler, latent updated in
 * place (pipelines/distilled.py:214-253; scripts/generate.py:1942-1979).  x0_out may be NULL.  */
int ltx2_dit_denoise_step(ltx2_dit* ctx, float* latent, const float* timesteps, int n_timesteps, const float* sigma_dev,
                          const float* mask, const float* clean, float sigma, float sigma_next, float* x0_out, void* stream);
/* Joint audio+video step (pipelines/distilled.py:198-271): both latents updated in place;
 * sigma_dev = device copy of sigma.                                                          */
int ltx2_dit_denoise_step_av(ltx2_dit* ctx, float* v_latent, float* a_latent, const float* v_timesteps,
                             int n_v_timesteps, const float* a_timesteps, int n_a_timesteps, const float* sigma_dev,
                             const float* v_mask, const float* v_clean, const float* a_mask, const float* a_clean,
                             float sigma, float sigma_next, float* v_x0_out, float* a_x0_out, void* stream);

/* hipGraph: capture n_steps of ltx2_dit_denoise_step over host_sigmas[n_steps+1] with a uniform
 * sigma per step (timesteps = sigma for every token), then replay.  latent is updated in place. */
int ltx2_dit_graph_capture(ltx2_dit* ctx, float* latent, const float* host_sigmas, int n_steps, void* stream);
/* The same for CONDITIONED loops (image-to-video; reference pipelines/common.py:193-232, pipelines/distilled.py:214-253): mask fp32 [N] = the
 * denoise mask (1 = free token), clean fp32 [N][C] = the clean latent of the conditioned tokens.  Step i runs with per-token timesteps
 * mask * sigma_i (formed on the device inside the captured step) and blends x0 with `clean` before the Euler update, exactly as
 * ltx2_dit_denoise_step with n_timesteps = N, mask and clean.  The workspace must be bound with per_token = 1.  A null mask (per modality in
 * the _av form) = no conditioning tokens there.  n_mask / n_clean: ELEMENT counts of the two buffers, checked against the bound token count
 * and N * out_channels (the replay reads exactly that much on every step).  (round 4; the counts: round 5, ABI 3)                       */
int ltx2_dit_graph_capture_cond(ltx2_dit* ctx, float* latent, const float* host_sigmas, int n_steps, const float* mask, int64_t n_mask, const float* clean,
                                int64_t n_clean, void* stream);
int ltx2_dit_graph_capture_cond_av(ltx2_dit* ctx, float* v_latent, float* a_latent, const float* host_sigmas, int n_steps, const float* v_mask, int64_t n_v_mask,
                                   const float* v_clean, int64_t n_v_clean, const float* a_mask, int64_t n_a_mask, const float* a_clean, int64_t n_a_clean, void* stream);
int ltx2_dit_graph_capture_av(ltx2_dit* ctx, float* v_latent, float* a_latent, const float* host_sigmas, int n_steps,
                              void* stream);
int ltx2_dit_graph_launch(ltx2_dit* ctx, void* stream);

/* Text-context key mask of one modality (0 = video, 1 = audio): Modality.context_mask as the reference's boolean (B, S) mask
 * (model.py:163-201 turns it into the additive -finfo.max mask of attention.py:38-70): mask fp32 [S] on the device, non-zero = the key may
 * be attended; NULL clears it.  Applies to the text cross-attention of every block until changed; call after binding the workspace
 * (binding clears it).  Every pipeline of the reference passes context_mask = None (pipelines/common.py:223-232).                  */
int ltx2_dit_set_context_mask(ltx2_dit* ctx, int modality, const float* mask, int S, void* stream);

/* Engine options, by name (unknown names -> LTX2_E_INVALID):
 *   "fp8_compute" = 1 (before ltx2_dit_bind_workspace): every linear of the VIDEO stream whose weight is registered fp8-resident (LTX2_DTYPE_FP8_E4M3FN codes +
 *   `<name>_scale`) and whose GEMM has M >= 1024 rows runs as ltx2_gemm_fp8 on activations quantised per token
 *   (ltx2_quantize_rows_fp8).  Default 0: fp8-resident weights are expanded to bf16 inside the GEMM (bit-identical to the
 *   reference's dequantise-at-load).
 *   "adaln_combine" = 0 (any time): tables and timestep embeddings reach every kernel separately, as in round 3.  Default 1: with one
 *   timestep per modality the sums of all layers are formed by one launch at the top of the step (bit-identical results).
 *   "fold_norms" = 0 / 1 (any time; default 1; VideoOnly non-V2.3 models on dense 16-bit weights in the bfloat16 build, one timestep per modality):
 *   1: the text cross-attention's plain RMS pre-norm rides on attn1.to_out's epilogue and attn2.to_q's accumulators (ltx2_gemm_bf16_fold) instead of
 *   running as a pass over the residual stream; 0: round 5's form.  Same mathematics; the operand is rounded before the row factor instead of after it.
 *   "text_kv_ahead" = 0 (any time; AudioVideo models with cross_attention_adaln only): the video stream projects its sigma-modulated text K / V
 *   inline, in front of its text cross-attention (round 4's schedule).  Default 1 (round 5): they are a function of the prompt and sigma only,
 *   so layer l's are projected on the side stream at the top of layer l, beside the main stream's norm / QKV projection (bit-identical
 *   results; -1.4 ... -1.8 ms per LTX-2.3 step).  Ignored with fp8_compute (that projection shares the video stream's activation scratch).  */
int ltx2_dit_set_option(ltx2_dit* ctx, const char* name, int value);


/* Measurement aid: bracket every launch of one GEMM kernel instantiation (epilogue id, or -1 for
 * every GEMM) issued by this thread's ltx2_dit_* calls with HIP events on the launch stream;
 * profile_end synchronises and returns the summed kernel time, launch count and 2*M*N*K flops. */
int ltx2_dit_profile_begin(ltx2_dit* ctx, int epilogue);
int ltx2_dit_profile_end(ltx2_dit* ctx, double* total_ms, int64_t* launches, double* flops);

/* ------------------------------------------------------------------------------------------
 * VAE decoder engine: SimpleVideoDecoder.__call__ (model/video_vae/simple_decoder.py:446-563)
 * ------------------------------------------------------------------------------------------ */
typedef struct ltx2_vae ltx2_vae;

#define LTX2_VAE_MAX_BLOCKS 16
#define LTX2_VAE_RES 0
#define LTX2_VAE_UPSAMPLE 1

typedef struct ltx2_vae_config {
    int n_blocks;                             /* up_blocks in execution order (reversed decoder_blocks) */
    int kind[LTX2_VAE_MAX_BLOCKS];            /* LTX2_VAE_RES | LTX2_VAE_UPSAMPLE */
    int num_layers[LTX2_VAE_MAX_BLOCKS];      /* res: number of ResBlock3d */
    int stride[LTX2_VAE_MAX_BLOCKS][3];       /* upsample: (ft, fh, fw) */
    int multiplier[LTX2_VAE_MAX_BLOCKS];      /* upsample: out_channels_reduction_factor */
    int residual[LTX2_VAE_MAX_BLOCKS];
    int base_channels;                        /* 128 -> first feature width 1024 */
    int latent_channels;                      /* 128 */
    int timestep_conditioning;
    float decode_noise_scale;                 /* 0.025 */
} ltx2_vae_config;

int ltx2_vae_create(const ltx2_vae_config* cfg, ltx2_vae** out);
void ltx2_vae_destroy(ltx2_vae* ctx);
/* Names = checkpoint keys (simple_decoder.py:592-671).  conv weights: bf16 [Cout][27][Cin]
 * (upsample convs row-permuted for the depth-to-space epilogue); linear weights bf16 [out,in];
 * everything else fp32.                                                                       */
int ltx2_vae_set_weight(ltx2_vae* ctx, const char* name, const void* ptr, int dtype, int64_t numel);
/* value of the checkpoint scalar vae.decoder.timestep_scale_multiplier (default 1000) */
int ltx2_vae_set_timestep_multiplier(ltx2_vae* ctx, float multiplier);
int64_t ltx2_vae_workspace_bytes(const ltx2_vae* ctx, int T, int H, int W);
int ltx2_vae_bind_workspace(ltx2_vae* ctx, void* ptr, int64_t bytes);
/* latent fp32 [C][T][H][W] -> video fp32 [3][To][32H][32W] in [-1, 1].  timestep < 0 disables
 * timestep conditioning for this call; noise (fp32, latent-shaped, N(0,1)) may be NULL (zeros). */
int ltx2_vae_decode(ltx2_vae* ctx, const float* latent, int T, int H, int W, float timestep, const float* noise,
                    int causal, float* video, void* stream);
/* output frame count of one decode call for T latent frames */
int ltx2_vae_out_frames(const ltx2_vae* ctx, int T);

#ifdef __cplusplus
}
#endif
#endif /* LTX2HIP_H */
