"""Loop and conditioning helpers shared by the pipelines.  Mirrors reference LTX_2_MLX/pipelines/common.py:23-262."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from ..conditioning.latent import VideoConditionByLatentIndex
from ..conditioning.tools import VideoLatentTools
from ..model.transformer import Modality
from ..types import LatentState


@dataclass
class ImageCondition:
    """An image that replaces the latent at one latent frame (reference pipelines/common.py:23-29).  `image` may
    carry an already loaded tensor (1, 3, 1, H, W) in [-1, 1] instead of a path."""
    image_path: Optional[str]
    frame_index: int
    strength: float = 0.95
    image: Optional[torch.Tensor] = None


def load_image_tensor(image_path: str, height: int, width: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """File -> (1, 3, 1, H, W) in [-1, 1]: RGB, aspect-preserving LANCZOS resize + centre crop
    (reference pipelines/common.py:32-102)."""
    import numpy as np
    from PIL import Image
    if not os.path.exists(image_path):
        raise FileNotFoundError(f"Image not found: {image_path}")
    try:
        img = Image.open(image_path)
    except Exception as e:  # noqa: BLE001
        raise ValueError(f"Failed to open image {image_path}: {e}")
    if img.mode not in ("RGB", "RGBA", "L"):
        raise ValueError(f"Unsupported image format: {img.mode}. Supported formats: RGB, RGBA, L")
    img = img.convert("RGB")
    sw, sh = img.size
    if abs(sw / sh - width / height) < 0.01:
        img = img.resize((width, height), Image.Resampling.LANCZOS)
    else:
        if sw / sh > width / height:
            nh, nw = height, int(sw * (height / sh))
        else:
            nw, nh = width, int(sh * (width / sw))
        img = img.resize((nw, nh), Image.Resampling.LANCZOS)
        left, top = (nw - width) // 2, (nh - height) // 2
        img = img.crop((left, top, left + width, top + height))
    arr = torch.from_numpy(np.array(img).astype(np.float32) / 127.5 - 1.0)
    return arr.permute(2, 0, 1)[None, :, None].to(dtype)


def create_image_conditionings(images: List[ImageCondition], video_encoder, height: int, width: int,
                               dtype: torch.dtype = torch.float32) -> List[VideoConditionByLatentIndex]:
    """Encode every image with the VAE encoder into a latent-index conditioning (reference pipelines/common.py:105-146)."""
    out = []
    for ic in images:
        img = ic.image if ic.image is not None else load_image_tensor(ic.image_path, height, width, dtype)
        if tuple(img.shape[-2:]) != (height, width):
            raise ValueError(f"image is {tuple(img.shape[-2:])}, expected ({height}, {width})")
        if video_encoder is None:
            raise ValueError("image conditioning needs a video_encoder")
        out.append(VideoConditionByLatentIndex(latent=video_encoder(img), strength=ic.strength, latent_idx=ic.frame_index))
    return out


def apply_conditionings(latent_state: LatentState, conditionings, video_tools: VideoLatentTools) -> LatentState:
    for c in conditionings:
        latent_state = c.apply_to(latent_state, video_tools)
    return latent_state


def post_process_latent(denoised: torch.Tensor, denoise_mask: torch.Tensor, clean_latent: torch.Tensor) -> torch.Tensor:
    """Blend denoised output with the clean state: denoised*mask + clean*(1-mask)."""
    if denoise_mask.ndim == 2 and denoised.ndim == 3:
        denoise_mask = denoise_mask[..., None]
    return (denoised * denoise_mask + clean_latent * (1 - denoise_mask)).to(denoised.dtype)


def timesteps_from_mask(denoise_mask: torch.Tensor, sigma: float) -> torch.Tensor:
    return denoise_mask * sigma


def modality_from_state(state: LatentState, context: torch.Tensor, sigma: float, enabled: bool = True, uniform: bool = False) -> Modality:
    """uniform=True: the caller KNOWS the denoise mask is all ones (checked once per loop, not per step), so the timesteps are
    the scalar sigma -- the broadcast AdaLN path, identical arithmetic, N x fewer AdaLN MLP rows."""
    # context_mask is always None (reference pipelines/common.py:223-232)
    ts = torch.full((1,), float(sigma), device=state.latent.device) if uniform else timesteps_from_mask(state.denoise_mask, sigma)
    return Modality(enabled=enabled, latent=state.latent, timesteps=ts,
                    positions=state.positions, context=context, context_mask=None,
                    sigma=torch.tensor([sigma], device=state.latent.device))


def audio_modality_from_state(state: LatentState, context: torch.Tensor, sigma: float, enabled: bool = True, uniform: bool = False) -> Modality:
    """Same record for the audio modality (reference pipelines/common.py:235-262)."""
    return modality_from_state(state, context, sigma, enabled, uniform)
