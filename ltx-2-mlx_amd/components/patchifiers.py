"""(B,C,F,H,W) <-> (B,N,C) layout changes and the position table (host glue on torch tensors).
Mirrors reference LTX_2_MLX/components/patchifiers.py:36-411 (video patch size 1, audio)."""
from __future__ import annotations

import math
from typing import Tuple

import torch

from ..types import AudioLatentShape, SpatioTemporalScaleFactors, VideoLatentShape


class VideoLatentPatchifier:
    def __init__(self, patch_size: int = 1):
        if patch_size != 1:
            raise ValueError("only patch_size=1 is used by LTX-2 (reference pipelines/distilled.py:139)")
        self._patch_size = (1, patch_size, patch_size)

    @property
    def patch_size(self) -> Tuple[int, int, int]:
        return self._patch_size

    def get_token_count(self, tgt_shape: VideoLatentShape) -> int:
        return tgt_shape.frames * tgt_shape.height * tgt_shape.width // math.prod(self._patch_size)

    def patchify(self, latents: torch.Tensor) -> torch.Tensor:
        b, c, f, h, w = latents.shape
        return latents.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)

    def unpatchify(self, latents: torch.Tensor, output_shape: VideoLatentShape) -> torch.Tensor:
        b = latents.shape[0]
        c, f, h, w = output_shape.channels, output_shape.frames, output_shape.height, output_shape.width
        return latents.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)

    def get_patch_grid_bounds(self, output_shape: VideoLatentShape, device=None) -> torch.Tensor:
        """[batch, 3, N, 2] start/end bounds in latent grid units (patchifiers.py:147-199)."""
        f, h, w = output_shape.frames, output_shape.height, output_shape.width
        gf, gh, gw = torch.meshgrid(torch.arange(f), torch.arange(h), torch.arange(w), indexing="ij")
        starts = torch.stack([gf, gh, gw], dim=0).reshape(3, -1)
        coords = torch.stack([starts, starts + 1], dim=-1)
        out = coords[None].expand(output_shape.batch, -1, -1, -1)
        return out.to(device) if device is not None else out


def get_pixel_coords(latent_coords: torch.Tensor, scale_factors: SpatioTemporalScaleFactors,
                     causal_fix: bool = False) -> torch.Tensor:
    """Scale to pixel space; causal fix shifts/clamps the temporal axis (patchifiers.py:202-240)."""
    scale = torch.tensor([scale_factors.time, scale_factors.height, scale_factors.width],
                         device=latent_coords.device).reshape(1, 3, 1, 1)
    px = latent_coords * scale
    if causal_fix:
        t = torch.clamp(px[:, 0] + 1 - scale_factors.time, min=0)
        px = torch.cat([t[:, None], px[:, 1:]], dim=1)
    return px


class AudioPatchifier:
    """(B, C, T, F) <-> (B, T, C*F) and causal [start, end) seconds per audio latent frame
    (reference components/patchifiers.py:243-411)."""

    def __init__(self, patch_size: int, sample_rate: int = 16000, hop_length: int = 160, audio_latent_downsample_factor: int = 4,
                 is_causal: bool = True, shift: int = 0):
        self.hop_length, self.sample_rate = hop_length, sample_rate
        self.audio_latent_downsample_factor = audio_latent_downsample_factor
        self.is_causal, self.shift = is_causal, shift
        self._patch_size = (1, patch_size, patch_size)

    @property
    def patch_size(self) -> Tuple[int, int, int]:
        return self._patch_size

    def get_token_count(self, tgt_shape: AudioLatentShape) -> int:
        return tgt_shape.frames

    def _get_audio_latent_time_in_sec(self, start_latent: int, end_latent: int) -> torch.Tensor:
        mel = torch.arange(start_latent, end_latent, dtype=torch.float32) * self.audio_latent_downsample_factor
        if self.is_causal:
            mel = torch.clamp(mel + 1 - self.audio_latent_downsample_factor, min=0)
        return mel * self.hop_length / self.sample_rate

    def patchify(self, audio_latents: torch.Tensor) -> torch.Tensor:
        b, c, t, f = audio_latents.shape
        return audio_latents.permute(0, 2, 1, 3).reshape(b, t, c * f)

    def unpatchify(self, audio_latents: torch.Tensor, output_shape: AudioLatentShape) -> torch.Tensor:
        b, t, _ = audio_latents.shape
        return audio_latents.reshape(b, t, output_shape.channels, output_shape.mel_bins).permute(0, 2, 1, 3)

    def get_patch_grid_bounds(self, output_shape: AudioLatentShape, device=None) -> torch.Tensor:
        """[batch, 1, T, 2] start/end timestamps in seconds (patchifiers.py:314-347,398-411)."""
        n = output_shape.frames
        t = torch.stack([self._get_audio_latent_time_in_sec(self.shift, n + self.shift),
                         self._get_audio_latent_time_in_sec(self.shift + 1, n + self.shift + 1)], dim=-1)
        out = t[None, None].expand(output_shape.batch, 1, n, 2).contiguous()
        return out.to(device) if device is not None else out
