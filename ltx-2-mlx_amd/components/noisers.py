"""Gaussian noiser (torch RNG; the reference's MLX Threefry stream is not reproducible, so
parity is defined on supplied noise).  Mirrors reference LTX_2_MLX/components/noisers.py:18-78."""
from __future__ import annotations

from typing import Optional

import torch

from ..types import LatentState


class GaussianNoiser:
    def __init__(self, generator: Optional[torch.Generator] = None):
        self.generator = generator

    def __call__(self, latent_state: LatentState, noise_scale: float = 1.0,
                 noise: Optional[torch.Tensor] = None) -> LatentState:
        lat = latent_state.latent
        if noise is None:
            noise = torch.randn(lat.shape, generator=self.generator, device=lat.device, dtype=lat.dtype)
        mask = latent_state.denoise_mask
        sm = (mask[..., None] if mask.ndim == 2 else mask) * noise_scale
        return latent_state.replace(latent=(noise * sm + lat * (1 - sm)).to(lat.dtype))
