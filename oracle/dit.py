"""Oracle: LTX-2 DiT (video-only V1) velocity / x0 model, PyTorch fp32 on CPU.

Test infrastructure (see oracle/__init__.py).  Restates, function by
function, the arithmetic of the reference's transformer path.  All citations
are ``path:line`` under /root/reference.

Weights are a flat ``dict[str, Tensor]`` keyed by the *checkpoint* names the
reference loader consumes after stripping ``model.diffusion_model.``
(LTX_2_MLX/loader/weight_converter.py:277-315): e.g.
``transformer_blocks.3.attn1.to_q.weight``, ``transformer_blocks.3.ff.net.0.proj.weight``,
``adaln_single.emb.timestep_embedder.linear_1.weight``.  Linear weights are
``[out, in]`` (no transposes, weight_converter.py:303-307).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class DiTConfig:
    """Constructor arguments of LTXModel (model/transformer/model.py:436-461)."""

    num_attention_heads: int = 32
    attention_head_dim: int = 128
    in_channels: int = 128
    out_channels: int = 128
    num_layers: int = 48
    caption_channels: Optional[int] = 3840
    norm_eps: float = 1e-6
    positional_embedding_theta: float = 10000.0
    positional_embedding_max_pos: List[int] = field(default_factory=lambda: [20, 2048, 2048])
    timestep_scale_multiplier: float = 1000.0

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def linear(x: Tensor, w: Dict[str, Tensor], name: str) -> Tensor:
    b = w.get(name + ".bias")
    return F.linear(x, w[name + ".weight"].float(), None if b is None else b.float())


# ---------------------------------------------------------------------------
# Norms  (attention.py:88-112, transformer.py:16-31, model.py:553)
# ---------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Optional[Tensor] = None, eps: float = 1e-6) -> Tensor:
    """mx.fast.rms_norm: x * rsqrt(mean(x^2, -1) + eps) [* weight]  (attention.py:100)."""
    y = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y if weight is None else y * weight.float()


def adaln_forward(x: Tensor, scale: Tensor, shift: Tensor, eps: float) -> Tensor:
    """_compiled_adaln_forward: rms_norm(x) * (1 + scale) + shift  (transformer.py:16-31)."""
    return rms_norm(x, None, eps) * (1.0 + scale) + shift


# ---------------------------------------------------------------------------
# Timestep embedding / AdaLN-single  (timestep_embedding.py:10-60,89-124,166-202)
# ---------------------------------------------------------------------------
def sinusoidal_timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0
    (timestep_embedding.py:36-54, Timesteps defaults :141-145) -> [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / float(half)
    arg = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def adaln_single(t_scaled: Tensor, w: Dict[str, Tensor], prefix: str) -> Tuple[Tensor, Tensor]:
    """AdaLayerNormSingle.__call__ (timestep_embedding.py:187-202).

    Returns (emb [T, n*D], embedded_timestep [T, D])."""
    e = sinusoidal_timestep_embedding(t_scaled, 256)
    e = linear(e, w, prefix + ".emb.timestep_embedder.linear_1")
    e = F.silu(e)
    e = linear(e, w, prefix + ".emb.timestep_embedder.linear_2")
    emb = linear(F.silu(e), w, prefix + ".linear")
    return emb, e


# ---------------------------------------------------------------------------
# RoPE (SPLIT)  (rope.py:92-144,181-211,214-289,292-328,365-418)
# ---------------------------------------------------------------------------
def rope_freq_grid(theta: float, n_pos_dims: int, inner_dim: int) -> Tensor:
    """generate_freq_grid (rope.py:181-211): theta**linspace(0,1,inner_dim//(2*n_dims)) * pi/2."""
    n = inner_dim // (2 * n_pos_dims)
    lin = torch.linspace(0.0, 1.0, n, dtype=torch.float32)
    return (torch.tensor(float(theta)) ** lin * (math.pi / 2)).float()


def rope_split_tables(
    positions: Tensor, dim: int, heads: int, theta: float, max_pos: List[int]
) -> Tuple[Tensor, Tensor]:
    """precompute_freqs_cis(rope_type=SPLIT, use_middle_indices_grid=True)
    (rope.py:365-418).  positions: [B, n_dims, T, 2] (start, end).
    Returns cos, sin of shape [B, H, T, dim // (2*H)] fp32."""
    n_dims = positions.shape[1]
    assert n_dims == len(max_pos)
    grid = rope_freq_grid(theta, n_dims, dim)                      # [n_freq]
    mid = (positions[..., 0] + positions[..., 1]) / 2.0            # rope.py:261-266  [B, n_dims, T]
    frac = torch.stack([mid[:, i, :] / max_pos[i] for i in range(n_dims)], dim=-1)  # :228-239 [B,T,n_dims]
    scaled = frac * 2 - 1                                          # :276
    freqs = grid[None, None, None, :] * scaled[..., None]          # [B, T, n_dims, n_freq]  :283
    freqs = freqs.transpose(2, 3).reshape(freqs.shape[0], freqs.shape[1], -1)  # slot = f*n_dims + d  :285-287
    cos, sin = torch.cos(freqs), torch.sin(freqs)
    pad = dim // 2 - freqs.shape[-1]                               # :409-411
    if pad:
        cos = torch.cat([torch.ones_like(cos[..., :pad]), cos], dim=-1)   # pad at the FRONT  :311-317
        sin = torch.cat([torch.zeros_like(sin[..., :pad]), sin], dim=-1)
    b, t, _ = cos.shape
    cos = cos.reshape(b, t, heads, -1).transpose(1, 2)             # head h <- slots [h*d/2, (h+1)*d/2)  :320-326
    sin = sin.reshape(b, t, heads, -1).transpose(1, 2)
    return cos.contiguous(), sin.contiguous()


def apply_split_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_split_rotary_emb (rope.py:92-144). x: [B, T, H*d]; cos/sin: [B, H, T, d/2]."""
    b, h, t, half = cos.shape
    xh = x.reshape(b, t, h, 2, half).permute(0, 2, 1, 3, 4)       # [B,H,T,2,d/2]
    first, second = xh[..., 0, :], xh[..., 1, :]
    o1 = first * cos - second * sin
    o2 = second * cos + first * sin
    out = torch.stack([o1, o2], dim=-2).reshape(b, h, t, 2 * half)
    return out.permute(0, 2, 1, 3).reshape(b, t, h * 2 * half)


# ---------------------------------------------------------------------------
# Attention  (attention.py:12-34,203-253)
# ---------------------------------------------------------------------------
def prepare_attention_mask(context_mask: Optional[Tensor]) -> Optional[Tensor]:
    """LTXModel._prepare_attention_mask (model.py:163-201) for fp32 compute: a boolean / integer key mask (B, S) becomes the additive
    mask (1 - m) * -3.40e38 of shape (B, 1, 1, S); a float mask is taken as it is (already additive)."""
    if context_mask is None:
        return None
    if context_mask.dtype in (torch.float16, torch.float32, torch.bfloat16, torch.float64):
        return context_mask
    m = (1 - context_mask.float()) * -3.40e38
    return m.reshape(context_mask.shape[0], 1, 1, context_mask.shape[-1])


def sdpa(q: Tensor, k: Tensor, v: Tensor, heads: int, mask: Optional[Tensor] = None) -> Tensor:
    """_compiled_attention_core_no_mask / _with_mask: softmax(q k^T / sqrt(d) + mask) v per head (attention.py:21-34, 38-70)."""
    b, tq, hd = q.shape
    tk = k.shape[1]
    d = hd // heads
    qh = q.reshape(b, tq, heads, d).transpose(1, 2)
    kh = k.reshape(b, tk, heads, d).transpose(1, 2)
    vh = v.reshape(b, tk, heads, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * (1.0 / math.sqrt(d))
    if mask is not None:
        if mask.ndim == 2:
            mask = mask[None, None]
        elif mask.ndim == 3:
            mask = mask[:, None]
        s = s + mask.to(s.dtype)
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vh)
    return o.transpose(1, 2).reshape(b, tq, hd)


def attention(
    x: Tensor,
    w: Dict[str, Tensor],
    prefix: str,
    heads: int,
    eps: float,
    context: Optional[Tensor] = None,
    pe: Optional[Tuple[Tensor, Tensor]] = None,
    mask: Optional[Tensor] = None,
) -> Tensor:
    """Attention.__call__ (attention.py:203-253): to_q/k/v (+bias), RMSNorm(weight)
    over the FULL inner dim on q and k (:186-187,231-232), SPLIT RoPE on q,k if pe,
    SDPA (additive mask when given), to_out."""
    ctx = x if context is None else context
    q = linear(x, w, prefix + ".to_q")
    k = linear(ctx, w, prefix + ".to_k")
    v = linear(ctx, w, prefix + ".to_v")
    q = rms_norm(q, w[prefix + ".q_norm.weight"], eps)
    k = rms_norm(k, w[prefix + ".k_norm.weight"], eps)
    if pe is not None:
        q = apply_split_rope(q, pe[0], pe[1])
        k = apply_split_rope(k, pe[0], pe[1])
    o = sdpa(q, k, v, heads, mask)
    return linear(o, w, prefix + ".to_out.0")


# ---------------------------------------------------------------------------
# Transformer block  (transformer.py:191-238 == video half of :503-529,622-631)
# ---------------------------------------------------------------------------
def feed_forward(x: Tensor, w: Dict[str, Tensor], prefix: str) -> Tensor:
    """FeedForward: Linear(D->4D) -> GELU(tanh) -> Linear(4D->D), NOT gated (feed_forward.py:29-54)."""
    h = F.gelu(linear(x, w, prefix + ".net.0.proj"), approximate="tanh")
    return linear(h, w, prefix + ".net.2")


def transformer_block(
    x: Tensor,
    context: Tensor,
    timesteps: Tensor,
    pe: Tuple[Tensor, Tensor],
    w: Dict[str, Tensor],
    i: int,
    cfg: DiTConfig,
    context_mask: Optional[Tensor] = None,
) -> Tensor:
    """BasicTransformerBlock.__call__ (transformer.py:191-238).
    timesteps: [B, T, 6, D] with T in {1, N}; AdaLN row order (shift, scale, gate) (:207-209)."""
    p = f"transformer_blocks.{i}"
    table = w[p + ".scale_shift_table"].float()                   # [6, D]
    ada = table[None, None, :, :] + timesteps                      # transformer.py:182-189
    shift_msa, scale_msa, gate_msa = ada[:, :, 0], ada[:, :, 1], ada[:, :, 2]
    h = adaln_forward(x, scale_msa, shift_msa, cfg.norm_eps)
    a = attention(h, w, p + ".attn1", cfg.num_attention_heads, cfg.norm_eps, pe=pe)
    x = x + a * gate_msa                                           # _compiled_residual_gate :35-46
    c = attention(rms_norm(x, None, cfg.norm_eps), w, p + ".attn2", cfg.num_attention_heads,
                  cfg.norm_eps, context=context, mask=context_mask)   # :217-226 (no RoPE; additive key mask when the Modality carries one)
    x = x + c
    shift_mlp, scale_mlp, gate_mlp = ada[:, :, 3], ada[:, :, 4], ada[:, :, 5]
    h = adaln_forward(x, scale_mlp, shift_mlp, cfg.norm_eps)
    f = feed_forward(h, w, p + ".ff")
    return x + f * gate_mlp


# ---------------------------------------------------------------------------
# Model  (model.py:113-161,203-281,744-758,776-881,895-936)
# ---------------------------------------------------------------------------
def caption_projection(context: Tensor, w: Dict[str, Tensor]) -> Tensor:
    """PixArtAlphaTextProjection (model.py:52-56)."""
    h = F.gelu(linear(context, w, "caption_projection.linear_1"), approximate="tanh")
    return linear(h, w, "caption_projection.linear_2")


def prepare_timestep(timesteps: Tensor, w: Dict[str, Tensor], cfg: DiTConfig, batch: int) -> Tuple[Tensor, Tensor]:
    """TransformerArgsPreprocessor._prepare_timestep (model.py:113-140)."""
    t = timesteps.float() * cfg.timestep_scale_multiplier
    emb, e = adaln_single(t.flatten(), w, "adaln_single")
    d = cfg.inner_dim
    return emb.reshape(batch, -1, emb.shape[-1] // d, d), e.reshape(batch, -1, d)


def velocity_model(
    latent: Tensor,
    context: Tensor,
    timesteps: Tensor,
    positions: Tensor,
    w: Dict[str, Tensor],
    cfg: DiTConfig,
    return_hidden: bool = False,
    context_mask: Optional[Tensor] = None,
):
    """LTXModel.__call__ for VideoOnly (model.py:776-881).

    latent [B,N,128], context [B,S,C_ctx], timesteps [B] or [B,N] / [B,N,1],
    positions [B,3,N,2].  Returns velocity [B,N,128] fp32."""
    b = latent.shape[0]
    x = linear(latent.float(), w, "patchify_proj")                # model.py:242
    ts = timesteps.reshape(b, -1)
    emb, e = prepare_timestep(ts, w, cfg, b)                       # emb [B,T,6,D], e [B,T,D]
    ctx = context.float()
    if cfg.caption_channels is not None:
        ctx = caption_projection(ctx, w)                           # model.py:142-161
    ctx = ctx.reshape(b, -1, cfg.inner_dim)
    pe = rope_split_tables(positions.float(), cfg.inner_dim, cfg.num_attention_heads,
                           cfg.positional_embedding_theta, cfg.positional_embedding_max_pos)
    hidden = []
    amask = prepare_attention_mask(context_mask)                     # model.py:266-268
    for i in range(cfg.num_layers):
        x = transformer_block(x, ctx, emb, pe, w, i, cfg, amask)
        if return_hidden:
            hidden.append(x)
    # _process_video_output (model.py:744-758): rows (shift, scale); LayerNorm no affine
    ss = w["scale_shift_table"].float()[None, None, :, :] + e[:, :, None, :]
    shift, scale = ss[:, :, 0], ss[:, :, 1]
    x = F.layer_norm(x, (cfg.inner_dim,), eps=cfg.norm_eps)
    x = x * (1 + scale) + shift
    v = linear(x, w, "proj_out")
    return (v, hidden) if return_hidden else v


def x0_model(latent: Tensor, context: Tensor, timesteps: Tensor, positions: Tensor,
             w: Dict[str, Tensor], cfg: DiTConfig, context_mask: Optional[Tensor] = None) -> Tensor:
    """X0Model.__call__ (model.py:895-936): x0 = latent - sigma * velocity."""
    v = velocity_model(latent, context, timesteps, positions, w, cfg, context_mask=context_mask)
    t = timesteps.float()
    if t.ndim == 1:
        t = t[:, None, None]
    elif t.ndim == 2:
        t = t[:, :, None]
    return latent.float() - t * v


# ---------------------------------------------------------------------------
# Synthetic weights (N(0, 0.02) weights; no checkpoints exist here)
# ---------------------------------------------------------------------------
def dit_weight_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    d = cfg.inner_dim
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    lin("patchify_proj", d, cfg.in_channels)
    lin("adaln_single.emb.timestep_embedder.linear_1", d, 256)
    lin("adaln_single.emb.timestep_embedder.linear_2", d, d)
    lin("adaln_single.linear", 6 * d, d)
    if cfg.caption_channels is not None:
        lin("caption_projection.linear_1", d, cfg.caption_channels)
        lin("caption_projection.linear_2", d, d)
    s["scale_shift_table"] = (2, d)
    lin("proj_out", cfg.out_channels, d)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(f"{p}.{a}.{n}", d, d)
            s[f"{p}.{a}.q_norm.weight"] = (d,)
            s[f"{p}.{a}.k_norm.weight"] = (d,)
        lin(f"{p}.ff.net.0.proj", 4 * d, d)
        lin(f"{p}.ff.net.2", d, 4 * d)
        s[f"{p}.scale_shift_table"] = (6, d)
    return s


def make_dit_weights(cfg: DiTConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.02,
                     norm_jitter: float = 0.1) -> Dict[str, Tensor]:
    """Seeded synthetic weights, regenerated identically on both boxes (never shipped)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in dit_weight_shapes(cfg).items():
        if name.endswith("_norm.weight"):
            t = 1.0 + norm_jitter * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = bias_std * torch.randn(shape, generator=g)
        elif name.endswith("scale_shift_table"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        out[name] = t
    return out
