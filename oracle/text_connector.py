"""Oracle: text-side per-prompt path between Gemma's hidden states and the DiT, PyTorch fp32 on CPU.

Test infrastructure (see oracle/__init__.py).  Restates the reference's
LTX_2_MLX/model/text_encoder/{connector.py, feature_extractor.py} and the part of encoder.py that chains them
(`encode_projected` / `encode_from_hidden_states`, :136-215); citations are ``path:line`` under /root/reference.
Gemma itself is out of scope: inputs are its hidden states (or already-projected features).

Weights: flat ``dict[str, Tensor]`` keyed by the checkpoint names the reference loader reads
(encoder.py:441-520) with the prefix ``model.diffusion_model.video_embeddings_connector.`` stripped:
``learnable_registers``, ``transformer_1d_blocks.{i}.attn1.{to_q,to_k,to_v,to_out.0}.{weight,bias}``,
``transformer_1d_blocks.{i}.attn1.{q_norm,k_norm}.weight``, ``transformer_1d_blocks.{i}.ff.net.0.proj.*``,
``transformer_1d_blocks.{i}.ff.net.2.*``; feature extractor ``aggregate_embed.weight`` (V1, no bias) or
``video_aggregate_embed.* / audio_aggregate_embed.*`` (V2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import dit as D

Tensor = torch.Tensor


@dataclass
class ConnectorConfig:
    """Constructor arguments of Embeddings1DConnector (connector.py:114-126)."""

    attention_head_dim: int = 128
    num_attention_heads: int = 30
    num_layers: int = 2
    positional_embedding_theta: float = 10000.0
    positional_embedding_max_pos: List[int] = field(default_factory=lambda: [1])
    num_learnable_registers: Optional[int] = 128
    norm_eps: float = 1e-6
    apply_gated_attention: bool = False
    double_precision_rope: bool = False

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# ---------------------------------------------------------------------------
# INTERLEAVED RoPE on a 1-D index grid  (rope.py:41-89,147-211,242-289,330-362,365-418)
# ---------------------------------------------------------------------------
def rope_interleaved_tables(seq_len: int, dim: int, theta: float, max_pos: List[int], double_precision: bool = False) -> Tuple[Tensor, Tensor]:
    """precompute_freqs_cis(indices_grid=arange(T)[None,None,:], rope_type=INTERLEAVED) (connector.py:253-270):
    grid = theta**linspace(0,1,dim//2) * pi/2 (float64 power when double_precision, rope.py:147-178);
    freqs = grid * (2 * idx/max_pos - 1) (rope.py:228-239,276-283); cos/sin repeated pairwise (:347-350);
    dim % 2 leading (cos 1, sin 0) pad slots (:352-360).  Returns cos, sin [1, T, dim] fp32."""
    n = dim // 2
    if double_precision:
        grid = torch.from_numpy((np.power(float(theta), np.linspace(0.0, 1.0, n, dtype=np.float64)) * math.pi / 2).astype(np.float32))
    else:
        grid = D.rope_freq_grid(theta, 1, dim)
    idx = torch.arange(seq_len, dtype=torch.float32)
    scaled = idx / float(max_pos[0]) * 2 - 1
    freqs = scaled[:, None] * grid[None, :]                        # [T, n]
    cos = torch.cos(freqs).repeat_interleave(2, dim=-1)
    sin = torch.sin(freqs).repeat_interleave(2, dim=-1)
    pad = dim % 2
    if pad:
        cos = torch.cat([torch.ones(seq_len, pad), cos], dim=-1)
        sin = torch.cat([torch.zeros(seq_len, pad), sin], dim=-1)
    return cos[None].contiguous(), sin[None].contiguous()


def apply_interleaved_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_interleaved_rotary_emb (rope.py:41-89): pairs (d0,d1),(d2,d3),..: x*cos + (-x_odd, x_even)*sin."""
    xp = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    rot = torch.stack([-xp[..., 1], xp[..., 0]], dim=-1).reshape(x.shape)
    return x * cos + rot * sin


def connector_attention(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int, eps: float, pe: Tuple[Tensor, Tensor]) -> Tensor:
    """Attention.__call__ as used by BasicTransformerBlock1D (attention.py:203-253): q/k/v projections, RMSNorm
    (weight) over the full inner dim, INTERLEAVED RoPE on q and k, SDPA without a mask (the connector clears it
    when registers are appended, connector.py:222-228), optional per-head gates, to_out."""
    q = D.rms_norm(D.linear(x, w, prefix + ".to_q"), w[prefix + ".q_norm.weight"], eps)
    k = D.rms_norm(D.linear(x, w, prefix + ".to_k"), w[prefix + ".k_norm.weight"], eps)
    v = D.linear(x, w, prefix + ".to_v")
    q = apply_interleaved_rope(q, pe[0], pe[1])
    k = apply_interleaved_rope(k, pe[0], pe[1])
    o = D.sdpa(q, k, v, heads)
    if prefix + ".to_gate_logits.weight" in w:                      # attention.py:241-249
        b, t, hd = o.shape
        gates = 2.0 * torch.sigmoid(D.linear(x, w, prefix + ".to_gate_logits"))
        o = (o.reshape(b, t, heads, hd // heads) * gates[..., None]).reshape(b, t, hd)
    return D.linear(o, w, prefix + ".to_out.0")


def connector_block(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: ConnectorConfig, pe: Tuple[Tensor, Tensor]) -> Tensor:
    """BasicTransformerBlock1D.__call__ (connector.py:61-101): x += attn(rms(x)); x += ff(rms(x)); no AdaLN."""
    x = x + connector_attention(D.rms_norm(x, None, cfg.norm_eps), w, prefix + ".attn1", cfg.num_attention_heads, cfg.norm_eps, pe)
    return x + D.feed_forward(D.rms_norm(x, None, cfg.norm_eps), w, prefix + ".ff")


def append_learnable_registers(x: Tensor, registers: Tensor) -> Tensor:
    """_append_learnable_registers (connector.py:173-230): extend to max(1024, T) rounded up to a multiple of the
    register count with the tiled registers' rows [T:]; the original rows (pad tokens included) stay in place."""
    b, t, d = x.shape
    n = registers.shape[0]
    dup = math.ceil(max(1024, t) / n)
    extra = registers.float().repeat(dup, 1)[t:]
    if extra.shape[0] == 0:
        return x
    return torch.cat([x, extra[None].expand(b, -1, -1)], dim=1)


def embeddings_connector(x: Tensor, w: Dict[str, Tensor], cfg: ConnectorConfig) -> Tensor:
    """Embeddings1DConnector.__call__ (connector.py:232-283) with registers (mask cleared): append registers,
    INTERLEAVED RoPE over arange(T'), the blocks, final weightless RMSNorm.  x [B, T, inner_dim] -> [B, T', inner_dim]."""
    if cfg.num_learnable_registers:
        x = append_learnable_registers(x, w["learnable_registers"])
    pe = rope_interleaved_tables(x.shape[1], cfg.inner_dim, cfg.positional_embedding_theta, cfg.positional_embedding_max_pos,
                                 cfg.double_precision_rope)
    for i in range(cfg.num_layers):
        x = connector_block(x, w, f"transformer_1d_blocks.{i}", cfg, pe)
    return D.rms_norm(x, None, cfg.norm_eps)


# ---------------------------------------------------------------------------
# Feature extractors  (feature_extractor.py:9-87,90-157,160-230)
# ---------------------------------------------------------------------------
def norm_and_concat_padded_batch(enc: Tensor, seq_lens: Tensor, padding_side: str = "right") -> Tensor:
    """feature_extractor.py:9-87.  enc [B,T,D,L]; per (batch, layer): masked mean over valid tokens x D, masked
    min/max; 8*(x-mean)/(range+eps); layers concatenated as [B,T,D*L] (index d*L + l); pad rows zeroed."""
    b, t, d, nl = enc.shape
    eps = 1e-6
    idx = torch.arange(t)[None, :]
    if padding_side == "right":
        mask = idx < seq_lens[:, None]
    elif padding_side == "left":
        mask = idx >= (t - seq_lens[:, None])
    else:
        raise ValueError(f"padding_side must be 'left' or 'right', got {padding_side}")
    m4 = mask[:, :, None, None]
    masked = torch.where(m4, enc, torch.zeros_like(enc))
    mean = masked.sum(dim=(1, 2), keepdim=True) / ((seq_lens * d).reshape(b, 1, 1, 1) + eps)
    x_min = torch.where(m4, enc, torch.full_like(enc, 1e9)).amin(dim=(1, 2), keepdim=True)
    x_max = torch.where(m4, enc, torch.full_like(enc, -1e9)).amax(dim=(1, 2), keepdim=True)
    normed = (8 * (enc - mean) / (x_max - x_min + eps)).reshape(b, t, d * nl)
    return torch.where(mask[:, :, None], normed, torch.zeros_like(normed))


def norm_and_concat_per_token_rms(enc: Tensor, attention_mask: Tensor) -> Tensor:
    """feature_extractor.py:160-181 (V2): per token and layer x * rsqrt(mean_D(x^2) + 1e-6); pad rows zeroed."""
    b, t, d, nl = enc.shape
    normed = (enc * torch.rsqrt(enc.pow(2).mean(dim=2, keepdim=True) + 1e-6)).reshape(b, t, d * nl)
    return torch.where(attention_mask.bool()[:, :, None], normed, torch.zeros_like(normed))


def feature_extractor_v1(hidden_states: List[Tensor], attention_mask: Tensor, w: Dict[str, Tensor], padding_side: str = "left") -> Tensor:
    """GemmaFeaturesExtractorProjLinear.extract_from_hidden_states (feature_extractor.py:125-157): stack the
    per-layer states on a trailing axis, normalise, Linear(D*L -> D, no bias)."""
    stacked = torch.stack([h.float() for h in hidden_states], dim=-1)
    seq_lens = attention_mask.sum(dim=-1).to(torch.int32)
    return D.linear(norm_and_concat_padded_batch(stacked, seq_lens, padding_side), w, "aggregate_embed")


def feature_extractor_v2(hidden_states: List[Tensor], attention_mask: Tensor, w: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """GemmaFeaturesExtractorV2.extract_from_hidden_states (feature_extractor.py:206-230): per-token RMS norm,
    then for each modality rescale by sqrt(target_dim / embedding_dim) and Linear(D*L -> target_dim) + bias."""
    stacked = torch.stack([h.float() for h in hidden_states], dim=-1)
    normed = norm_and_concat_per_token_rms(stacked, attention_mask)
    d = hidden_states[0].shape[-1]
    v_dim, a_dim = w["video_aggregate_embed.weight"].shape[0], w["audio_aggregate_embed.weight"].shape[0]
    video = D.linear(normed * math.sqrt(v_dim / d), w, "video_aggregate_embed")
    audio = D.linear(normed * math.sqrt(a_dim / d), w, "audio_aggregate_embed")
    return video, audio


def encode_projected(features: Tensor, attention_mask: Tensor, w: Dict[str, Tensor], cfg: ConnectorConfig) -> Tuple[Tensor, Tensor]:
    """VideoGemmaTextEncoderModel.encode_projected (encoder.py:183-215): connector, then the binary output mask
    (all ones once registers were appended) multiplies the encoding.  Returns (encoding [B,T',D], mask [B,T'])."""
    out = embeddings_connector(features.float(), w, cfg)
    if cfg.num_learnable_registers:
        mask = torch.ones(out.shape[0], out.shape[1], dtype=torch.int32)
    else:
        mask = attention_mask.to(torch.int32)
    return out * mask[:, :, None], mask


# ---------------------------------------------------------------------------
# Synthetic weights
# ---------------------------------------------------------------------------
def connector_weight_shapes(cfg: ConnectorConfig) -> Dict[str, Tuple[int, ...]]:
    d = cfg.inner_dim
    s: Dict[str, Tuple[int, ...]] = {}
    if cfg.num_learnable_registers:
        s["learnable_registers"] = (cfg.num_learnable_registers, d)
    for i in range(cfg.num_layers):
        p = f"transformer_1d_blocks.{i}"
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{p}.attn1.{n}.weight"] = (d, d)
            s[f"{p}.attn1.{n}.bias"] = (d,)
        s[f"{p}.attn1.q_norm.weight"] = (d,)
        s[f"{p}.attn1.k_norm.weight"] = (d,)
        if cfg.apply_gated_attention:
            s[f"{p}.attn1.to_gate_logits.weight"] = (cfg.num_attention_heads, d)
            s[f"{p}.attn1.to_gate_logits.bias"] = (cfg.num_attention_heads,)
        s[f"{p}.ff.net.0.proj.weight"] = (4 * d, d)
        s[f"{p}.ff.net.0.proj.bias"] = (4 * d,)
        s[f"{p}.ff.net.2.weight"] = (d, 4 * d)
        s[f"{p}.ff.net.2.bias"] = (d,)
    return s


def make_connector_weights(cfg: ConnectorConfig, seed: int = 0, std: float = 0.02) -> Dict[str, Tensor]:
    """Seeded synthetic weights: N(0, std) matrices and biases, norm weights 1 + N(0, 0.1), registers U(-1, 1)
    (the reference's initialisation, connector.py:165-171)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}
    for k, shp in connector_weight_shapes(cfg).items():
        if k == "learnable_registers":
            w[k] = torch.rand(shp, generator=g) * 2 - 1
        elif k.endswith("_norm.weight"):
            w[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            w[k] = std * torch.randn(shp, generator=g)
    return w
