"""Euler stepping through the C ABI (ltx2_euler_step).  Mirrors reference
LTX_2_MLX/components/diffusion_steps.py:24-67 and core_utils.py:34-94."""
from __future__ import annotations

from typing import Union

import torch

from .. import kernels as K


def _sig(s) -> float:
    return float(s.item()) if isinstance(s, torch.Tensor) else float(s)


def to_velocity(sample: torch.Tensor, sigma: Union[float, torch.Tensor], denoised_sample: torch.Tensor) -> torch.Tensor:
    s = _sig(sigma)
    if s == 0:
        raise ValueError("Sigma can't be 0.0")
    return ((sample.float() - denoised_sample.float()) / s).to(sample.dtype)


def to_denoised(sample: torch.Tensor, velocity: torch.Tensor, sigma: Union[float, torch.Tensor]) -> torch.Tensor:
    return (sample.float() - velocity.float() * (sigma.float() if isinstance(sigma, torch.Tensor) else sigma)).to(sample.dtype)


class EulerDiffusionStep:
    """sample + (sample - denoised)/sigma * (sigma_next - sigma), computed in fp32 on the GPU."""

    def step(self, sample: torch.Tensor, denoised_sample: torch.Tensor, sigmas, step_index: int) -> torch.Tensor:
        sigma, sigma_next = _sig(sigmas[step_index]), _sig(sigmas[step_index + 1])
        shp = sample.shape
        c = shp[-1]
        out = K.euler_step(sample.float().reshape(-1, c), denoised_sample.float().reshape(-1, c), sigma, sigma_next)
        return out.reshape(shp).to(sample.dtype)
