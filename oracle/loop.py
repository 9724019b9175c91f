"""Oracle: sampling-loop helpers around the DiT (host-side scalar / layout code).

Test infrastructure (see oracle/__init__.py).  Citations are ``path:line``
under /root/reference.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor

# components/schedulers.py:236-253 (asserted by reference tests/test_scheduler.py:103-144)
DISTILLED_SIGMA_VALUES = [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
STAGE_2_DISTILLED_SIGMA_VALUES = [0.909375, 0.725, 0.421875, 0.0]

BASE_SHIFT_ANCHOR = 1024
MAX_SHIFT_ANCHOR = 4096


def ltx2_scheduler(steps: int, tokens: Optional[int] = None, max_shift: float = 2.05, base_shift: float = 0.95,
                   stretch: bool = True, terminal: float = 0.1) -> Tensor:
    """LTX2Scheduler.execute (components/schedulers.py:30-102)."""
    if tokens is None:
        tokens = MAX_SHIFT_ANCHOR
    sig = torch.linspace(1.0, 0.0, steps + 1, dtype=torch.float32)
    mm = (max_shift - base_shift) / (MAX_SHIFT_ANCHOR - BASE_SHIFT_ANCHOR)
    b = base_shift - mm * BASE_SHIFT_ANCHOR
    e = math.exp(tokens * mm + b)
    safe = torch.where(sig != 0, sig, torch.ones_like(sig))
    sig = torch.where(sig != 0, e / (e + (1.0 / safe - 1.0)), torch.zeros_like(sig))
    if stretch and steps > 0:
        one_minus = 1.0 - sig
        scale = float(one_minus[steps - 1]) / (1.0 - terminal)
        stretched = 1.0 - one_minus / scale
        sig = torch.where(sig != 0, stretched, sig)
    return sig.float()


def latent_shape_from_pixels(frames: int, height: int, width: int) -> Tuple[int, int, int]:
    """VideoLatentShape.from_pixel_shape (types.py:72-87): F'=(F-1)//8+1, H//32, W//32."""
    return (frames - 1) // 8 + 1, height // 32, width // 32


def patchify(latent: Tensor) -> Tensor:
    """VideoLatentPatchifier.patchify, patch size 1 (components/patchifiers.py:74-102): (B,C,F,H,W)->(B,N,C)."""
    b, c, f, h, w = latent.shape
    return latent.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c)


def unpatchify(tokens: Tensor, f: int, h: int, w: int) -> Tensor:
    """VideoLatentPatchifier.unpatchify (components/patchifiers.py:104-145)."""
    b, n, c = tokens.shape
    return tokens.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)


def video_positions(batch: int, f: int, h: int, w: int, fps: float, causal_fix: bool = True) -> Tensor:
    """get_patch_grid_bounds + get_pixel_coords + seconds conversion
    (components/patchifiers.py:147-240; conditioning/tools.py:67-78; scripts/generate.py:1808-1826).
    Returns [B, 3, N, 2] fp32: (t in seconds, y px, x px) x [start, end)."""
    gf, gh, gw = torch.meshgrid(torch.arange(f), torch.arange(h), torch.arange(w), indexing="ij")
    starts = torch.stack([gf, gh, gw], dim=0).reshape(3, -1).float()
    coords = torch.stack([starts, starts + 1.0], dim=-1)                     # [3, N, 2]
    scale = torch.tensor([8.0, 32.0, 32.0]).reshape(3, 1, 1)
    px = coords * scale
    if causal_fix:
        px[0] = torch.clamp(px[0] + 1 - 8, min=0)                            # patchifiers.py:227-238
    px[0] = px[0] / fps
    return px[None].expand(batch, -1, -1, -1).contiguous()


def to_velocity(sample: Tensor, sigma: float, denoised: Tensor) -> Tensor:
    """core_utils.to_velocity (core_utils.py:34-63)."""
    if sigma == 0:
        raise ValueError("Sigma can't be 0.0")
    return (sample.float() - denoised.float()) / sigma


def euler_step(sample: Tensor, denoised: Tensor, sigma: float, sigma_next: float) -> Tensor:
    """EulerDiffusionStep.step / euler_step_x0 (components/diffusion_steps.py:36-67; scripts/generate.py:905-930)."""
    v = to_velocity(sample, sigma, denoised)
    return sample.float() + v * float(sigma_next - sigma)


def post_process_latent(denoised: Tensor, denoise_mask: Tensor, clean: Tensor) -> Tensor:
    """pipelines/common.py:169-190."""
    if denoise_mask.ndim == 2 and denoised.ndim == 3:
        denoise_mask = denoise_mask[..., None]
    return denoised * denoise_mask + clean * (1 - denoise_mask)


def timesteps_from_mask(denoise_mask: Tensor, sigma: float) -> Tensor:
    """pipelines/common.py:193-203."""
    return denoise_mask * sigma


def gaussian_noiser(latent: Tensor, denoise_mask: Tensor, noise: Tensor, noise_scale: float) -> Tensor:
    """GaussianNoiser.__call__ with a SUPPLIED noise tensor (components/noisers.py:36-78)."""
    m = denoise_mask[..., None] if denoise_mask.ndim == 2 else denoise_mask
    sm = m * noise_scale
    return noise * sm + latent * (1 - sm)


def denoise_loop_cli(latent: Tensor, x0_fn: Callable[[Tensor, float], Tensor], sigmas: Sequence[float]) -> Tensor:
    """The standard CLI loop (scripts/generate.py:1797-1979): latent (B,C,F,H,W);
    x0_fn(tokens [B,N,C], sigma) -> x0 tokens; timesteps shape (B,)."""
    b, c, f, h, w = latent.shape
    x = latent.float()
    for i in range(len(sigmas) - 1):
        x0 = unpatchify(x0_fn(patchify(x), float(sigmas[i])), f, h, w)
        x = euler_step(x, x0, float(sigmas[i]), float(sigmas[i + 1]))
    return x


def denoise_loop_pipeline(tokens: Tensor, denoise_mask: Tensor, clean: Tensor,
                          x0_fn: Callable[[Tensor, Tensor, float], Tensor], sigmas: Sequence[float]) -> Tensor:
    """DistilledPipeline._denoise_loop_av, video-only branch (pipelines/distilled.py:198-272):
    per-token timesteps = mask * sigma; x0_fn(tokens, timesteps [B,N,1], sigma)."""
    x = tokens.float()
    for i in range(len(sigmas) - 1):
        s = float(sigmas[i])
        x0 = x0_fn(x, timesteps_from_mask(denoise_mask, s), s)
        x0 = post_process_latent(x0, denoise_mask, clean)
        x = euler_step(x, x0, s, float(sigmas[i + 1]))
    return x
