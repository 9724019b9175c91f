"""Oracle: LTX-2 AudioVideo DiT (19B "V1" AV blocks and the 22B "V2.3" variant with 9-row AdaLN,
prompt-modulated text K/V and per-head attention gates), PyTorch fp32 on CPU.

Test infrastructure (see oracle/__init__.py).  Restates BasicAVTransformerBlock.__call__
(LTX_2_MLX/model/transformer/transformer.py:241-648), MultiModalTransformerArgsPreprocessor
(model/transformer/model.py:284-410), the AudioVideo LTXModel.__call__ (model.py:776-881) and
X0Model (model.py:895-936) on top of the video-only pieces in oracle/dit.py.  Weight names are the
checkpoint names after stripping ``model.diffusion_model.`` (module attribute paths; Linear
weights ``[out, in]``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import dit as D

Tensor = torch.Tensor


@dataclass
class AVConfig:
    """LTXModel(model_type=AudioVideo, ...) constructor arguments (model.py:436-461) plus the audio
    class constants (model.py:428-434)."""

    num_attention_heads: int = 32
    attention_head_dim: int = 128
    audio_heads: int = 32
    audio_head_dim: int = 64
    in_channels: int = 128
    out_channels: int = 128
    audio_in_channels: int = 128
    audio_out_channels: int = 128
    num_layers: int = 48
    caption_channels: Optional[int] = 3840        # None for V2.3 (feature extractor projects directly)
    norm_eps: float = 1e-6
    positional_embedding_theta: float = 10000.0
    positional_embedding_max_pos: List[int] = field(default_factory=lambda: [20, 2048, 2048])
    audio_max_pos: int = 20                       # AUDIO_CROSS_PE_MAX_POS (model.py:434)
    timestep_scale_multiplier: float = 1000.0
    av_ca_timestep_scale_multiplier: float = 1.0
    cross_attention_adaln: bool = False           # V2.3: 9 AdaLN rows + prompt_scale_shift_table
    apply_gated_attention: bool = False           # V2.3: to_gate_logits, out *= 2*sigmoid(.)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @property
    def audio_inner_dim(self) -> int:
        return self.audio_heads * self.audio_head_dim

    @property
    def adaln_rows(self) -> int:
        return 9 if self.cross_attention_adaln else 6


# ---------------------------------------------------------------------------
# Attention with separate K RoPE tables and optional per-head gates (attention.py:203-253)
# ---------------------------------------------------------------------------
def attention(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int, eps: float, context: Optional[Tensor] = None,
              pe: Optional[Tuple[Tensor, Tensor]] = None, k_pe: Optional[Tuple[Tensor, Tensor]] = None, mask: Optional[Tensor] = None) -> Tensor:
    ctx = x if context is None else context
    q = D.linear(x, w, prefix + ".to_q")
    k = D.linear(ctx, w, prefix + ".to_k")
    v = D.linear(ctx, w, prefix + ".to_v")
    q = D.rms_norm(q, w[prefix + ".q_norm.weight"], eps)
    k = D.rms_norm(k, w[prefix + ".k_norm.weight"], eps)
    if pe is not None:
        kp = pe if k_pe is None else k_pe
        q = D.apply_split_rope(q, pe[0], pe[1])
        k = D.apply_split_rope(k, kp[0], kp[1])
    o = D.sdpa(q, k, v, heads, mask)                                # mask: the additive (B, 1, 1, S) text-context mask (attention.py:38-70) or None
    if prefix + ".to_gate_logits.weight" in w:                     # attention.py:241-249
        gates = 2.0 * torch.sigmoid(D.linear(x, w, prefix + ".to_gate_logits"))
        b, t, hd = o.shape
        o = (o.reshape(b, t, heads, hd // heads) * gates[..., None]).reshape(b, t, hd)
    return D.linear(o, w, prefix + ".to_out.0")


def _ada(table: Tensor, ts: Tensor, start: int, end: int):
    """get_ada_values (transformer.py:369-392): table[start:end] + timestep[:, :, start:end]."""
    a = table.float()[None, None, start:end, :] + ts[:, :, start:end, :]
    return tuple(a[:, :, i, :] for i in range(end - start))


def _text_cross(x, context, w, prefix, table, prompt_table, ts, prompt_ts, heads, cfg: AVConfig, mask: Optional[Tensor] = None):
    """_apply_text_cross_attention (transformer.py:427-455); mask = the modality's prepared context mask (model.py:163-201)."""
    if cfg.cross_attention_adaln:
        shift_q, scale_q, gate = _ada(table, ts, 6, 9)
        kv = prompt_table.float()[None, None, :, :] + prompt_ts
        shift_kv, scale_kv = kv[:, :, 0, :], kv[:, :, 1, :]
        h = D.rms_norm(x, None, cfg.norm_eps) * (1 + scale_q) + shift_q
        ehs = context * (1 + scale_kv) + shift_kv
        return attention(h, w, prefix, heads, cfg.norm_eps, context=ehs, mask=mask) * gate
    return attention(D.rms_norm(x, None, cfg.norm_eps), w, prefix, heads, cfg.norm_eps, context=context, mask=mask)


def av_block(vx: Tensor, ax: Tensor, va: dict, aa: dict, w: Dict[str, Tensor], i: int, cfg: AVConfig):
    """BasicAVTransformerBlock.__call__ with both modalities enabled and no perturbations
    (transformer.py:457-648).  `va` / `aa` carry the per-modality TransformerArgs fields."""
    p = f"transformer_blocks.{i}"
    eps = cfg.norm_eps
    Hv, Ha = cfg.num_attention_heads, cfg.audio_heads
    vt, at = w[p + ".scale_shift_table"], w[p + ".audio_scale_shift_table"]
    # video self-attention + text cross-attention (:503-529)
    sh, sc, g = _ada(vt, va["timesteps"], 0, 3)
    vx = vx + attention(D.adaln_forward(vx, sc, sh, eps), w, p + ".attn1", Hv, eps, pe=va["pe"]) * g
    vx = vx + _text_cross(vx, va["context"], w, p + ".attn2", vt, w.get(p + ".prompt_scale_shift_table"),
                          va["timesteps"], va.get("prompt_timestep"), Hv, cfg, va.get("context_mask"))
    # audio self-attention + text cross-attention (:531-554)
    sh, sc, g = _ada(at, aa["timesteps"], 0, 3)
    ax = ax + attention(D.adaln_forward(ax, sc, sh, eps), w, p + ".audio_attn1", Ha, eps, pe=aa["pe"]) * g
    ax = ax + _text_cross(ax, aa["context"], w, p + ".audio_attn2", at, w.get(p + ".audio_prompt_scale_shift_table"),
                          aa["timesteps"], aa.get("prompt_timestep"), Ha, cfg, aa.get("context_mask"))
    # audio <-> video cross-modal attention (:556-620); table rows (scale_a2v, shift_a2v, scale_v2a, shift_v2a, gate)
    vn, an = D.rms_norm(vx, None, eps), D.rms_norm(ax, None, eps)

    def ca(table, ss_ts, gate_ts):
        t = table.float()
        ss = t[None, None, :4, :] + ss_ts
        gt = t[None, None, 4:, :] + gate_ts
        return ss[:, :, 0], ss[:, :, 1], ss[:, :, 2], ss[:, :, 3], gt[:, :, 0]

    a_sc_a2v, a_sh_a2v, a_sc_v2a, a_sh_v2a, gate_v2a = ca(w[p + ".scale_shift_table_a2v_ca_audio"], aa["cross_ss"], aa["cross_gate"])
    v_sc_a2v, v_sh_a2v, v_sc_v2a, v_sh_v2a, gate_a2v = ca(w[p + ".scale_shift_table_a2v_ca_video"], va["cross_ss"], va["cross_gate"])
    vx = vx + attention(vn * (1 + v_sc_a2v) + v_sh_a2v, w, p + ".audio_to_video_attn", Ha, eps,
                        context=an * (1 + a_sc_a2v) + a_sh_a2v, pe=va["cross_pe"], k_pe=aa["cross_pe"]) * gate_a2v
    ax = ax + attention(an * (1 + a_sc_v2a) + a_sh_v2a, w, p + ".video_to_audio_attn", Ha, eps,
                        context=vn * (1 + v_sc_v2a) + v_sh_v2a, pe=aa["cross_pe"], k_pe=va["cross_pe"]) * gate_v2a
    # feed-forwards (:622-642)
    sh, sc, g = _ada(vt, va["timesteps"], 3, 6)
    vx = vx + D.feed_forward(D.adaln_forward(vx, sc, sh, eps), w, p + ".ff") * g
    sh, sc, g = _ada(at, aa["timesteps"], 3, 6)
    ax = ax + D.feed_forward(D.adaln_forward(ax, sc, sh, eps), w, p + ".audio_ff") * g
    return vx, ax


# ---------------------------------------------------------------------------
# Preprocessors (model.py:231-281, 284-410)
# ---------------------------------------------------------------------------
def _adaln(t_scaled: Tensor, w, prefix: str, batch: int, dim: int):
    emb, e = D.adaln_single(t_scaled.flatten(), w, prefix)
    return emb.reshape(batch, -1, emb.shape[-1] // dim, dim), e.reshape(batch, -1, dim)


def prepare_modality(latent, context, timesteps, sigma, positions, w, cfg: AVConfig, audio: bool, cross_sigma: Tensor,
                     context_mask: Optional[Tensor] = None) -> dict:
    """MultiModalTransformerArgsPreprocessor.prepare(modality, cross_modality) (model.py:366-410).  context_mask: the boolean / 0-1 (B, S) key
    mask of the text context, turned into the additive mask by D.prepare_attention_mask (model.py:163-201)."""
    pre = "audio_" if audio else ""
    dim = cfg.audio_inner_dim if audio else cfg.inner_dim
    heads = cfg.audio_heads if audio else cfg.num_attention_heads
    b = latent.shape[0]
    x = D.linear(latent.float(), w, pre + "patchify_proj")
    ts = timesteps.float().reshape(b, -1) * cfg.timestep_scale_multiplier
    emb, e = _adaln(ts, w, pre + "adaln_single", b, dim)
    out = {"x": x, "timesteps": emb, "embedded_timestep": e, "context_mask": D.prepare_attention_mask(context_mask)}
    if cfg.cross_attention_adaln:                                  # model.py:151-161
        s = sigma.float().reshape(b, -1)[:, 0] * cfg.timestep_scale_multiplier
        out["prompt_timestep"], _ = _adaln(s, w, pre + "prompt_adaln_single", b, dim)
    ctx = context.float()
    if cfg.caption_channels is not None:
        h = F.gelu(D.linear(ctx, w, pre + "caption_projection.linear_1"), approximate="tanh")
        ctx = D.linear(h, w, pre + "caption_projection.linear_2")
    out["context"] = ctx.reshape(b, -1, dim)
    max_pos = [cfg.audio_max_pos] if audio else cfg.positional_embedding_max_pos
    out["pe"] = D.rope_split_tables(positions.float(), dim, heads, cfg.positional_embedding_theta, max_pos)
    # cross-modal RoPE: temporal axis of THIS modality, audio inner dim, audio heads (model.py:320-344)
    out["cross_pe"] = D.rope_split_tables(positions.float()[:, 0:1], cfg.audio_inner_dim, heads,
                                          cfg.positional_embedding_theta, [cfg.audio_max_pos])
    # cross-attention timestep from the OTHER modality's sigma (model.py:346-364,392-404)
    cs = cross_sigma.float().reshape(b, -1)[:, 0] * cfg.timestep_scale_multiplier
    ss_name = "av_ca_audio_scale_shift_adaln_single" if audio else "av_ca_video_scale_shift_adaln_single"
    g_name = "av_ca_v2a_gate_adaln_single" if audio else "av_ca_a2v_gate_adaln_single"
    out["cross_ss"], _ = _adaln(cs, w, ss_name, b, dim)
    factor = cfg.av_ca_timestep_scale_multiplier / cfg.timestep_scale_multiplier
    out["cross_gate"], _ = _adaln(cs * factor, w, g_name, b, dim)
    return out


def _output(x, e, w, pre: str, dim: int, eps: float):
    """_process_video_output / _process_audio_output (model.py:744-774)."""
    ss = w[pre + "scale_shift_table"].float()[None, None, :, :] + e[:, :, None, :]
    x = F.layer_norm(x, (dim,), eps=eps) * (1 + ss[:, :, 1]) + ss[:, :, 0]
    return D.linear(x, w, pre + "proj_out")


def av_velocity_model(video: dict, audio: dict, w: Dict[str, Tensor], cfg: AVConfig, return_hidden: bool = False):
    """LTXModel.__call__ for AudioVideo with both modalities present (model.py:776-881).
    `video` / `audio`: dicts with latent [B,T,128], context, timesteps ([B] or [B,T,1]), sigma [B],
    positions ([B,3,N,2] / [B,1,Ta,2])."""
    vs = video.get("sigma", video["timesteps"])
    as_ = audio.get("sigma", audio["timesteps"])
    va = prepare_modality(video["latent"], video["context"], video["timesteps"], vs, video["positions"], w, cfg, False, as_, video.get("context_mask"))
    aa = prepare_modality(audio["latent"], audio["context"], audio["timesteps"], as_, audio["positions"], w, cfg, True, vs, audio.get("context_mask"))
    vx, ax = va["x"], aa["x"]
    hidden = []
    for i in range(cfg.num_layers):
        vx, ax = av_block(vx, ax, va, aa, w, i, cfg)
        if return_hidden:
            hidden.append((vx, ax))
    vv = _output(vx, va["embedded_timestep"], w, "", cfg.inner_dim, cfg.norm_eps)
    av = _output(ax, aa["embedded_timestep"], w, "audio_", cfg.audio_inner_dim, cfg.norm_eps)
    return (vv, av, hidden) if return_hidden else (vv, av)


def av_x0_model(video: dict, audio: dict, w, cfg: AVConfig):
    """X0Model.__call__ (model.py:895-936): x0 = latent - timesteps * velocity per modality."""
    vv, av = av_velocity_model(video, audio, w, cfg)

    def den(m, v):
        t = m["timesteps"].float()
        t = t[:, None, None] if t.ndim == 1 else (t[:, :, None] if t.ndim == 2 else t)
        return m["latent"].float() - t * v

    return den(video, vv), den(audio, av)


def video_only_x0_model(video: dict, w, cfg: AVConfig) -> Tensor:
    """X0Model(LTXModel) with NO audio tokens on the AudioVideo / V2.3 block program (model.py:829-840: video-only inference;
    transformer.py:479-483: run_ax and run_a2v are False for an empty audio stream, so a block is video self-attention, text
    cross-attention -- with the prompt AdaLN driven by Modality.sigma, model.py:151-158 -- and the video feed-forward).  This is
    also what a VideoOnly LTXModel with cross_attention_adaln / apply_gated_attention computes."""
    vs = video.get("sigma", video["timesteps"])
    va = prepare_modality(video["latent"], video["context"], video["timesteps"], vs, video["positions"], w, cfg, False, vs, video.get("context_mask"))
    x, eps, H = va["x"], cfg.norm_eps, cfg.num_attention_heads
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        vt = w[p + ".scale_shift_table"]
        sh, sc, g = _ada(vt, va["timesteps"], 0, 3)
        x = x + attention(D.adaln_forward(x, sc, sh, eps), w, p + ".attn1", H, eps, pe=va["pe"]) * g
        x = x + _text_cross(x, va["context"], w, p + ".attn2", vt, w.get(p + ".prompt_scale_shift_table"), va["timesteps"],
                            va.get("prompt_timestep"), H, cfg, va.get("context_mask"))
        sh, sc, g = _ada(vt, va["timesteps"], 3, 6)
        x = x + D.feed_forward(D.adaln_forward(x, sc, sh, eps), w, p + ".ff") * g
    v = _output(x, va["embedded_timestep"], w, "", cfg.inner_dim, eps)
    t = video["timesteps"].float()
    t = t[:, None, None] if t.ndim == 1 else (t[:, :, None] if t.ndim == 2 else t)
    return video["latent"].float() - t * v


def audio_positions(batch: int, num_steps: int, sample_rate: int = 16000, hop_length: int = 160, downsample: int = 4) -> Tensor:
    """AudioPatchifier.get_patch_grid_bounds (components/patchifiers.py:287-347,398-411): causal
    [start, end) seconds per audio latent frame -> [B, 1, T, 2]."""
    def sec(a, b):
        mel = torch.arange(a, b, dtype=torch.float32) * downsample
        mel = torch.clamp(mel + 1 - downsample, min=0)
        return mel * hop_length / sample_rate
    t = torch.stack([sec(0, num_steps), sec(1, num_steps + 1)], dim=-1)
    return t[None, None].expand(batch, 1, num_steps, 2).contiguous()


# ---------------------------------------------------------------------------
# Synthetic weights
# ---------------------------------------------------------------------------
def av_weight_shapes(cfg: AVConfig) -> Dict[str, Tuple[int, ...]]:
    dv, da = cfg.inner_dim, cfg.audio_inner_dim
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    def adaln(name, dim, n):
        lin(name + ".emb.timestep_embedder.linear_1", dim, 256)
        lin(name + ".emb.timestep_embedder.linear_2", dim, dim)
        lin(name + ".linear", n * dim, dim)

    def attn(name, qdim, cdim, inner, heads):
        lin(name + ".to_q", inner, qdim)
        lin(name + ".to_k", inner, cdim)
        lin(name + ".to_v", inner, cdim)
        lin(name + ".to_out.0", qdim, inner)
        s[name + ".q_norm.weight"] = (inner,)
        s[name + ".k_norm.weight"] = (inner,)
        if cfg.apply_gated_attention:
            lin(name + ".to_gate_logits", heads, qdim)

    for pre, dim, cin, cout in (("", dv, cfg.in_channels, cfg.out_channels), ("audio_", da, cfg.audio_in_channels, cfg.audio_out_channels)):
        lin(pre + "patchify_proj", dim, cin)
        adaln(pre + "adaln_single", dim, cfg.adaln_rows)
        if cfg.cross_attention_adaln:
            adaln(pre + "prompt_adaln_single", dim, 2)
        if cfg.caption_channels is not None:
            lin(pre + "caption_projection.linear_1", dim, cfg.caption_channels)
            lin(pre + "caption_projection.linear_2", dim, dim)
        s[pre + "scale_shift_table"] = (2, dim)
        lin(pre + "proj_out", cout, dim)
    adaln("av_ca_video_scale_shift_adaln_single", dv, 4)
    adaln("av_ca_a2v_gate_adaln_single", dv, 1)
    adaln("av_ca_audio_scale_shift_adaln_single", da, 4)
    adaln("av_ca_v2a_gate_adaln_single", da, 1)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        attn(p + ".attn1", dv, dv, dv, cfg.num_attention_heads)
        attn(p + ".attn2", dv, dv, dv, cfg.num_attention_heads)
        attn(p + ".audio_attn1", da, da, da, cfg.audio_heads)
        attn(p + ".audio_attn2", da, da, da, cfg.audio_heads)
        attn(p + ".audio_to_video_attn", dv, da, da, cfg.audio_heads)
        attn(p + ".video_to_audio_attn", da, dv, da, cfg.audio_heads)
        lin(p + ".ff.net.0.proj", 4 * dv, dv)
        lin(p + ".ff.net.2", dv, 4 * dv)
        lin(p + ".audio_ff.net.0.proj", 4 * da, da)
        lin(p + ".audio_ff.net.2", da, 4 * da)
        s[p + ".scale_shift_table"] = (cfg.adaln_rows, dv)
        s[p + ".audio_scale_shift_table"] = (cfg.adaln_rows, da)
        if cfg.cross_attention_adaln:
            s[p + ".prompt_scale_shift_table"] = (2, dv)
            s[p + ".audio_prompt_scale_shift_table"] = (2, da)
        s[p + ".scale_shift_table_a2v_ca_audio"] = (5, da)
        s[p + ".scale_shift_table_a2v_ca_video"] = (5, dv)
    return s


def make_av_weights(cfg: AVConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.02, norm_jitter: float = 0.1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in av_weight_shapes(cfg).items():
        if name.endswith("_norm.weight"):
            t = 1.0 + norm_jitter * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = bias_std * torch.randn(shape, generator=g)
        elif "scale_shift_table" in name:
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        out[name] = t
    return out
