"""Loop helpers shared by the pipelines.  Mirrors reference LTX_2_MLX/pipelines/common.py:169-232."""
from __future__ import annotations

import torch

from ..model.transformer import Modality
from ..types import LatentState


def post_process_latent(denoised: torch.Tensor, denoise_mask: torch.Tensor, clean_latent: torch.Tensor) -> torch.Tensor:
    """Blend denoised output with the clean state: denoised*mask + clean*(1-mask)."""
    if denoise_mask.ndim == 2 and denoised.ndim == 3:
        denoise_mask = denoise_mask[..., None]
    return (denoised * denoise_mask + clean_latent * (1 - denoise_mask)).to(denoised.dtype)


def timesteps_from_mask(denoise_mask: torch.Tensor, sigma: float) -> torch.Tensor:
    return denoise_mask * sigma


def modality_from_state(state: LatentState, context: torch.Tensor, sigma: float, enabled: bool = True) -> Modality:
    # context_mask is always None (reference pipelines/common.py:223-232)
    return Modality(enabled=enabled, latent=state.latent, timesteps=timesteps_from_mask(state.denoise_mask, sigma),
                    positions=state.positions, context=context, context_mask=None,
                    sigma=torch.tensor([sigma], device=state.latent.device))


def audio_modality_from_state(state: LatentState, context: torch.Tensor, sigma: float, enabled: bool = True) -> Modality:
    """Same record for the audio modality (reference pipelines/common.py:235-262)."""
    return modality_from_state(state, context, sigma, enabled)
