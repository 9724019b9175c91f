"""VideoLatentTools / AudioLatentTools: initial state, patchify/unpatchify, clear_conditioning.
Mirrors reference LTX_2_MLX/conditioning/tools.py:24-300."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ..components.patchifiers import AudioPatchifier, VideoLatentPatchifier, get_pixel_coords
from ..types import AudioLatentShape, LatentState, SpatioTemporalScaleFactors, VideoLatentShape

DEFAULT_SCALE_FACTORS = SpatioTemporalScaleFactors.default()


@dataclass(frozen=True)
class VideoLatentTools:
    patchifier: VideoLatentPatchifier
    target_shape: VideoLatentShape
    fps: float
    scale_factors: SpatioTemporalScaleFactors = DEFAULT_SCALE_FACTORS
    causal_fix: bool = True

    def create_initial_state(self, dtype: torch.dtype = torch.float32, initial_latent: Optional[torch.Tensor] = None,
                             device=None) -> LatentState:
        if initial_latent is not None:
            if tuple(initial_latent.shape) != self.target_shape.to_tuple():
                raise ValueError(f"Initial latent shape {tuple(initial_latent.shape)} does not match "
                                 f"target shape {self.target_shape.to_tuple()}")
            device = initial_latent.device
        else:
            initial_latent = torch.zeros(self.target_shape.to_tuple(), dtype=dtype, device=device)
        mask = torch.ones(self.target_shape.mask_shape().to_tuple(), dtype=torch.float32, device=device)
        coords = self.patchifier.get_patch_grid_bounds(self.target_shape, device=device)
        pos = get_pixel_coords(coords, self.scale_factors, causal_fix=self.causal_fix).float()
        pos = torch.cat([pos[:, 0:1] / self.fps, pos[:, 1:]], dim=1)
        return self.patchify(LatentState(latent=initial_latent, denoise_mask=mask, positions=pos, clean_latent=initial_latent))

    def patchify(self, s: LatentState) -> LatentState:
        p = self.patchifier.patchify
        return s.replace(latent=p(s.latent), denoise_mask=p(s.denoise_mask), clean_latent=p(s.clean_latent))

    def unpatchify(self, s: LatentState) -> LatentState:
        u = self.patchifier.unpatchify
        return s.replace(latent=u(s.latent, self.target_shape), clean_latent=u(s.clean_latent, self.target_shape),
                         denoise_mask=u(s.denoise_mask, self.target_shape.mask_shape()))

    def clear_conditioning(self, s: LatentState) -> LatentState:
        n = self.patchifier.get_token_count(self.target_shape)
        return LatentState(latent=s.latent[:, :n], clean_latent=s.clean_latent[:, :n],
                           denoise_mask=torch.ones_like(s.denoise_mask)[:, :n], positions=s.positions[:, :, :n])


@dataclass(frozen=True)
class AudioLatentTools:
    """Audio twin of VideoLatentTools (reference conditioning/tools.py:168-300): positions are the causal
    [start, end) seconds of every audio latent frame, shape (B, 1, T, 2)."""
    patchifier: AudioPatchifier
    target_shape: AudioLatentShape

    def create_initial_state(self, dtype: torch.dtype = torch.float32, initial_latent: Optional[torch.Tensor] = None,
                             device=None) -> LatentState:
        if initial_latent is not None:
            if tuple(initial_latent.shape) != self.target_shape.to_tuple():
                raise ValueError(f"Initial latent shape {tuple(initial_latent.shape)} does not match "
                                 f"target shape {self.target_shape.to_tuple()}")
            device = initial_latent.device
        else:
            initial_latent = torch.zeros(self.target_shape.to_tuple(), dtype=dtype, device=device)
        mask = torch.ones(self.target_shape.mask_shape().to_tuple(), dtype=torch.float32, device=device)
        pos = self.patchifier.get_patch_grid_bounds(self.target_shape, device=device).to(dtype)
        return self.patchify(LatentState(latent=initial_latent, denoise_mask=mask, positions=pos, clean_latent=initial_latent))

    def patchify(self, s: LatentState) -> LatentState:
        p = self.patchifier.patchify
        return s.replace(latent=p(s.latent), denoise_mask=p(s.denoise_mask), clean_latent=p(s.clean_latent))

    def unpatchify(self, s: LatentState) -> LatentState:
        u = self.patchifier.unpatchify
        return s.replace(latent=u(s.latent, self.target_shape), clean_latent=u(s.clean_latent, self.target_shape),
                         denoise_mask=u(s.denoise_mask, self.target_shape.mask_shape()))

    def clear_conditioning(self, s: LatentState) -> LatentState:
        n = self.patchifier.get_token_count(self.target_shape)
        return LatentState(latent=s.latent[:, :n], clean_latent=s.clean_latent[:, :n],
                           denoise_mask=torch.ones_like(s.denoise_mask)[:, :n], positions=s.positions[:, :, :n])
