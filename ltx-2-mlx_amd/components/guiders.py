"""Classifier-free guidance on the denoised prediction (reference LTX_2_MLX/components/guiders.py:26-77, 290-306).

Host-side glue of the guided single-stage loops: two x0 predictions (positive / negative prompt) per step are combined here in
fp32 on the device (element-wise ops and two dot products over a (B, N, C) tensor).  STG / APG guiders are outside the path."""
from __future__ import annotations

from dataclasses import dataclass

import torch


def projection_coef(to_project: torch.Tensor, project_onto: torch.Tensor) -> torch.Tensor:
    """<to_project, project_onto> / (|project_onto|^2 + 1e-8) per batch element, shape (B, 1) (guiders.py:290-306)."""
    b = to_project.shape[0]
    p, n = to_project.reshape(b, -1), project_onto.reshape(b, -1)
    return (p * n).sum(dim=1, keepdim=True) / ((n * n).sum(dim=1, keepdim=True) + 1e-8)


@dataclass(frozen=True)
class CFGGuider:
    """cond + (scale - 1) (cond - uncond)  (guiders.py:26-47)."""
    scale: float

    def delta(self, cond: torch.Tensor, uncond: torch.Tensor) -> torch.Tensor:
        return (self.scale - 1) * (cond - uncond)

    def guide(self, cond: torch.Tensor, uncond: torch.Tensor) -> torch.Tensor:
        return cond + self.delta(cond, uncond)

    def enabled(self) -> bool:
        return self.scale != 1.0


@dataclass(frozen=True)
class CFGStarRescalingGuider:
    """CFG with the unconditioned sample rescaled onto the conditioned one first (guiders.py:51-76): the coefficient (B, 1) multiplies a
    (B, N, C) tensor exactly as the reference's broadcast does."""
    scale: float

    def delta(self, cond: torch.Tensor, uncond: torch.Tensor) -> torch.Tensor:
        rescaled_neg = projection_coef(cond, uncond) * uncond
        return (self.scale - 1) * (cond - rescaled_neg)

    def guide(self, cond: torch.Tensor, uncond: torch.Tensor) -> torch.Tensor:
        return cond + self.delta(cond, uncond)

    def enabled(self) -> bool:
        return self.scale != 1.0
