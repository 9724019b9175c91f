"""Sigma schedules (host scalars).  Mirrors reference LTX_2_MLX/components/schedulers.py:30-102,236-278."""
from __future__ import annotations

import math
from typing import Optional

import torch

BASE_SHIFT_ANCHOR = 1024
MAX_SHIFT_ANCHOR = 4096

# Official distilled schedule: 9 values -> 8 steps; stage 2: 4 values -> 3 steps (schedulers.py:236-253)
DISTILLED_SIGMA_VALUES = [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
STAGE_2_DISTILLED_SIGMA_VALUES = [0.909375, 0.725, 0.421875, 0.0]


class LTX2Scheduler:
    """Token-count-shifted schedule stretched to a terminal value (schedulers.py:30-102)."""

    def execute(self, steps: int, latent: Optional[torch.Tensor] = None, max_shift: float = 2.05,
                base_shift: float = 0.95, stretch: bool = True, terminal: float = 0.1, **_kw) -> torch.Tensor:
        tokens = math.prod(latent.shape[2:]) if latent is not None else MAX_SHIFT_ANCHOR
        sig = torch.linspace(1.0, 0.0, steps + 1, dtype=torch.float32)
        mm = (max_shift - base_shift) / (MAX_SHIFT_ANCHOR - BASE_SHIFT_ANCHOR)
        shift = tokens * mm + (base_shift - mm * BASE_SHIFT_ANCHOR)
        e = math.exp(shift)
        nz = sig != 0
        safe = torch.where(nz, sig, torch.ones_like(sig))
        sig = torch.where(nz, e / (e + (1.0 / safe - 1.0)), torch.zeros_like(sig))
        if stretch and steps > 0:
            one_minus = 1.0 - sig
            scale = float(one_minus[steps - 1]) / (1.0 - terminal)
            sig = torch.where(sig != 0, 1.0 - one_minus / scale, sig)
        return sig.float()


def get_sigma_schedule(num_steps: int, distilled: bool = False, latent: Optional[torch.Tensor] = None) -> torch.Tensor:
    if distilled:
        return torch.tensor(DISTILLED_SIGMA_VALUES, dtype=torch.float32)
    return LTX2Scheduler().execute(steps=num_steps, latent=latent)
