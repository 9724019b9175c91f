"""(B,C,F,H,W) <-> (B,N,C) layout changes and the position table (host glue on torch tensors).
Mirrors reference LTX_2_MLX/components/patchifiers.py:36-240 (video half, patch size 1)."""
from __future__ import annotations

import math
from typing import Tuple

import torch

from ..types import SpatioTemporalScaleFactors, VideoLatentShape


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
