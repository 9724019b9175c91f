"""Gemma feature extractors behind the reference's API (LTX_2_MLX/model/text_encoder/feature_extractor.py:9-87
norm_and_concat_padded_batch, :90-157 GemmaFeaturesExtractorProjLinear, :160-230 per-token RMS + V2 extractor).

The 49-layer stack [B, T, 3840, 49] is never materialised in the reference's interleaved order: the projection
weight's input columns are re-ordered once at load time from (d * L + l) to (l * D + d), so the normalised states of
layer l are simply columns [l*D, (l+1)*D) of the GEMM's A operand, and the 188 160-wide contraction runs on
`ltx2_gemm_bf16`.  The per-(prompt, layer) statistics of the V1 normalisation are torch reductions (per prompt,
outside the metric); the V2 per-token RMS normalisation is `ltx2_adaln_rmsnorm`."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple, Union

import torch

from ... import kernels as K

BF16 = torch.bfloat16


def _valid_mask(t: int, sequence_lengths: torch.Tensor, padding_side: str) -> torch.Tensor:
    idx = torch.arange(t, device=sequence_lengths.device)[None, :]
    if padding_side == "right":
        return idx < sequence_lengths[:, None]
    if padding_side == "left":
        return idx >= (t - sequence_lengths[:, None])
    raise ValueError(f"padding_side must be 'left' or 'right', got {padding_side}")


def norm_and_concat_padded_batch(encoded_text: torch.Tensor, sequence_lengths: torch.Tensor, padding_side: str = "right") -> torch.Tensor:
    """[B, T, D, L] -> [B, T, D*L] (index d*L + l), 8*(x - masked mean)/(masked range + 1e-6) per (batch, layer), pad rows
    zero (feature_extractor.py:9-87).  Reference-order convenience form; the extractors below use the layer-major one."""
    b, t, d, nl = encoded_text.shape
    x = encoded_text.float()
    mask = _valid_mask(t, sequence_lengths, padding_side)
    m4 = mask[:, :, None, None]
    mean = torch.where(m4, x, torch.zeros_like(x)).sum(dim=(1, 2), keepdim=True) / ((sequence_lengths * d).reshape(b, 1, 1, 1) + 1e-6)
    x_min = torch.where(m4, x, torch.full_like(x, 1e9)).amin(dim=(1, 2), keepdim=True)
    x_max = torch.where(m4, x, torch.full_like(x, -1e9)).amax(dim=(1, 2), keepdim=True)
    normed = (8 * (x - mean) / (x_max - x_min + 1e-6)).reshape(b, t, d * nl)
    return torch.where(mask[:, :, None], normed, torch.zeros_like(normed))


def _layer_major(weight: torch.Tensor, hidden_dim: int, num_layers: int) -> torch.Tensor:
    """[out, D*L] with input index d*L + l  ->  [out, L*D] with input index l*D + d."""
    return weight.reshape(weight.shape[0], hidden_dim, num_layers).permute(0, 2, 1).reshape(weight.shape[0], -1)


class GemmaFeaturesExtractorProjLinear:
    """hidden states of all Gemma layers -> [B, T, hidden_dim] (V1; `text_embedding_projection.aggregate_embed.weight`,
    no bias)."""

    def __init__(self, hidden_dim: int = 3840, num_layers: int = 49, device: Union[str, torch.device] = "cuda"):
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GemmaFeaturesExtractorProjLinear runs on the MI355X only (no CPU fallback)")
        self._w = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        w = sd["aggregate_embed.weight"]
        if tuple(w.shape) != (self.hidden_dim, self.hidden_dim * self.num_layers):
            raise ValueError(f"aggregate_embed.weight: shape {tuple(w.shape)} != {(self.hidden_dim, self.hidden_dim * self.num_layers)}")
        self._w = _layer_major(w.to(self.device, torch.float32), self.hidden_dim, self.num_layers).to(BF16).contiguous()

    def extract_from_hidden_states(self, hidden_states: List[torch.Tensor], attention_mask: torch.Tensor, padding_side: str = "left") -> torch.Tensor:
        if self._w is None:
            raise RuntimeError("GemmaFeaturesExtractorProjLinear: weights not loaded")
        if len(hidden_states) != self.num_layers:
            raise ValueError(f"expected {self.num_layers} hidden states, got {len(hidden_states)}")
        b, t, d = hidden_states[0].shape
        am = attention_mask.to(self.device)
        seq = am.sum(dim=-1).to(torch.int32)
        mask = _valid_mask(t, seq, padding_side)
        denom = (seq * d).float() + 1e-6
        a = torch.empty(b, t, self.num_layers * d, device=self.device, dtype=BF16)
        for l, hs in enumerate(hidden_states):                       # per (batch, layer) masked mean / min / max
            x = hs.to(self.device, torch.float32)
            m3 = mask[:, :, None]
            mean = torch.where(m3, x, torch.zeros_like(x)).sum(dim=(1, 2)) / denom
            lo = torch.where(m3, x, torch.full_like(x, 1e9)).amin(dim=(1, 2))
            hi = torch.where(m3, x, torch.full_like(x, -1e9)).amax(dim=(1, 2))
            nrm = 8 * (x - mean[:, None, None]) / (hi - lo + 1e-6)[:, None, None]
            a[:, :, l * d:(l + 1) * d] = torch.where(m3, nrm, torch.zeros_like(nrm)).to(BF16)
        out = K.gemm(a.reshape(b * t, -1), self._w, None, epilogue=K.nv.EPI_F32)
        return out.reshape(b, t, self.hidden_dim)


class GemmaFeaturesExtractorV2:
    """V2.3: per-token RMS normalisation and two projections with bias, to the video and audio transformer widths."""

    def __init__(self, hidden_dim: int = 3840, num_layers: int = 49, video_inner_dim: int = 4096, audio_inner_dim: int = 2048,
                 device: Union[str, torch.device] = "cuda"):
        self.hidden_dim, self.num_layers, self.embedding_dim = hidden_dim, num_layers, hidden_dim
        self.video_inner_dim, self.audio_inner_dim = video_inner_dim, audio_inner_dim
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GemmaFeaturesExtractorV2 runs on the MI355X only (no CPU fallback)")
        self._wv = self._bv = self._wa = self._ba = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        flat = self.hidden_dim * self.num_layers
        for name, n in (("video_aggregate_embed", self.video_inner_dim), ("audio_aggregate_embed", self.audio_inner_dim)):
            if tuple(sd[name + ".weight"].shape) != (n, flat) or tuple(sd[name + ".bias"].shape) != (n,):
                raise ValueError(f"{name}: unexpected shape {tuple(sd[name + '.weight'].shape)}")
        # the sqrt(target_dim / embedding_dim) input rescale (feature_extractor.py:224-229) is folded into the weights
        def prep(name, n):
            w = sd[name + ".weight"].to(self.device, torch.float32) * math.sqrt(n / self.embedding_dim)
            return _layer_major(w, self.hidden_dim, self.num_layers).to(BF16).contiguous(), sd[name + ".bias"].to(self.device, torch.float32).contiguous()
        self._wv, self._bv = prep("video_aggregate_embed", self.video_inner_dim)
        self._wa, self._ba = prep("audio_aggregate_embed", self.audio_inner_dim)

    def extract_from_hidden_states(self, hidden_states: List[torch.Tensor], attention_mask: torch.Tensor, padding_side: str = "left") -> Tuple[torch.Tensor, torch.Tensor]:
        if self._wv is None:
            raise RuntimeError("GemmaFeaturesExtractorV2: weights not loaded")
        if len(hidden_states) != self.num_layers:
            raise ValueError(f"expected {self.num_layers} hidden states, got {len(hidden_states)}")
        b, t, d = hidden_states[0].shape
        valid = attention_mask.to(self.device).bool().reshape(b * t, 1)
        a = torch.empty(b * t, self.num_layers * d, device=self.device, dtype=BF16)
        for l, hs in enumerate(hidden_states):                       # x * rsqrt(mean_D(x^2) + 1e-6) per token and layer
            a[:, l * d:(l + 1) * d] = K.adaln_rmsnorm(hs.to(self.device, torch.float32).reshape(b * t, d), 1e-6)
        a = torch.where(valid, a, torch.zeros_like(a))              # pad rows contribute the bias only
        video = K.gemm(a, self._wv, self._bv, epilogue=K.nv.EPI_F32).reshape(b, t, self.video_inner_dim)
        audio = K.gemm(a, self._wa, self._ba, epilogue=K.nv.EPI_F32).reshape(b, t, self.audio_inner_dim)
        return video, audio
