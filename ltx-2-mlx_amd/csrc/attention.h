// Flash-attention forward (head_dim 128 / 64) + V^T re-layout: parameter block and host launchers.
#pragma once
#include "common.h"

struct AttnParams {
    const bf16* Q;   // [Nq][ldq], head h at columns h*128
    const bf16* K;   // [Nkv][ldk], head h at columns h*128
    const bf16* VT;  // [H][head_dim][Npad], key-permuted (see vt_transpose_launch)
    bf16* O;         // [Nq][ldo]
    long ldq, ldk, ldo, vt_head_stride;
    int Nq, Nkv, Npad, H;
    int head_dim;       // 128 (default when 0) or 64
    float scale_log2e;  // (1/sqrt(d)) * log2(e)
    // Per-query-row softmax scale (text cross-attention with q_norm folded in, round 3): q_ss[q * q_ss_ld + j], j < q_ss_ld, are partial sums
    // of the row's squared norm over the FULL inner dim q_norm_dim; row q then uses scale_log2e * rsqrt(sum / q_norm_dim + q_eps) -- the
    // RMS normalisation of q as a positive per-row factor on its scores (the row maximum is taken on the raw scores: order-preserving).
    // null = one scale for every row.
    const float* q_ss;
    int q_ss_ld, q_norm_dim;
    float q_eps;
    // Key mask (reference attention.py:38-70 with the boolean (B, S) context mask of model.py:163-201): bit i of kmask[t] = key 64 t + i may be
    // attended; a masked key's score is replaced by -1e30 (the reference ADDS -3.4e38 to it: the same softmax, including the uniform
    // result over the masked keys of a row whose keys are all masked).  null = no mask.
    const unsigned long long* kmask;
    // Per-head output gates (V2.3 apply_gated_attention, attention.py:241-249): out[q, h] *= 2 * sigmoid(gate[q * gate_ld + h]), applied to the fp32
    // result before it is rounded (round 4: was a pass over the attention output).  null = none.
    const float* gate;
    int gate_ld;
    // gate_parts > 1: `gate` holds that many partial logit arrays of [Nq][gate_ld] each (gate_logits_parts_launch) and gate_bias[H] is added to their sum
    int gate_parts;
    const float* gate_bias;
};

int attn_launch(const AttnParams& p, hipStream_t stream);
int vt_transpose_launch(const bf16* V, long ld, bf16* VT, int Nkv, int Npad, int H, hipStream_t stream, int head_dim = 128);
