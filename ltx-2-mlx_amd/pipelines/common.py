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
    sg = torch.full((1,), float(sigma), device=state.latent.device)
    ts = sg if uniform else timesteps_from_mask(state.denoise_mask, sigma)
    return Modality(enabled=enabled, latent=state.latent, timesteps=ts,
                    positions=state.positions, context=context, context_mask=None, sigma=sg)


def audio_modality_from_state(state: LatentState, context: torch.Tensor, sigma: float, enabled: bool = True, uniform: bool = False) -> Modality:
    """Same record for the audio modality (reference pipelines/common.py:235-262)."""
    return modality_from_state(state, context, sigma, enabled, uniform)


_eager_noted = set()


def _note_eager_once(why: str) -> None:
    """use_hip_graph=True but the loop cannot be replayed from the captured graph (say so once per reason)."""
    if why not in _eager_noted:
        _eager_noted.add(why)
        import sys
        print(f"  note: hipGraph replay skipped, the sampling loop runs eagerly: {why}", file=sys.stderr)


def joint_denoise_loop(transformer, is_av_model: bool, video_state: LatentState, audio_state: Optional[LatentState], sigmas,
                       video_context: torch.Tensor, audio_context: Optional[torch.Tensor], stepper, callback=None,
                       use_hip_graph: bool = False, *, negative_video_context: Optional[torch.Tensor] = None,
                       negative_audio_context: Optional[torch.Tensor] = None, video_guider=None, audio_guider=None):
    """The guidance-free sampling loop every pipeline here shares (reference pipelines/distilled.py:198-271 and the
    `need_cfg == False` branch of pipelines/one_stage.py:466-568 / :224-330).  Per step and modality:
    Modality(timesteps = denoise_mask * sigma) -> X0Model -> post_process_latent -> EulerDiffusionStep.
    Two equivalent executions: (a) API-faithful, one X0Model call + stepper.step per step; (b) hipGraph replay of the fused
    steps (ltx2_dit_graph_*) when the denoise masks are uniform and no callback needs intermediate states.
    `transformer` is an X0Model; returns (video_state, audio_state)."""
    sig = [float(s) for s in sigmas]
    n = len(sig) - 1
    model = transformer.velocity_model
    joint = is_av_model and audio_state is not None
    if joint and audio_context is None:
        raise ValueError("AudioVideo model: audio_encoding (the audio text context) is required")
    if is_av_model and not joint:
        model = model._video_twin()        # no audio tokens: the video half alone (reference model.py:829-840), same weights
    states = [video_state] + ([audio_state] if joint else [])
    uniform = all(bool((st.denoise_mask == 1).all()) for st in states)        # no conditioning tokens
    # classifier-free guidance (pipelines/one_stage.py:224-330 video, :466-568 joint): a second evaluation with the negative prompt whenever a
    # guider is enabled(); BOTH modalities are then guided by their own guider (a guider at scale 1 returns cond).  The negative prompt runs
    # through a second engine context over the same weights, so each context keeps its own per-prompt text K / V.
    need_cfg = (video_guider is not None and video_guider.enabled()) or (joint and audio_guider is not None and audio_guider.enabled())
    neg = None
    if need_cfg:
        if negative_video_context is None or (joint and negative_audio_context is None):
            raise ValueError("guidance needs the negative prompt's encoding(s)")
        from ..model.transformer import X0Model
        neg = X0Model(model.clone_sharing_weights())
    if use_hip_graph and not (callback is None and not need_cfg):
        _note_eager_once("a step callback" if callback is not None else "classifier-free guidance (two evaluations per step)")
    if use_hip_graph and callback is None and not need_cfg:
        lat = video_state.latent[0].float().contiguous()
        alat = audio_state.latent[0].float().contiguous() if joint else None
        # conditioning tokens (image-to-video): the captured steps form timesteps = denoise_mask * sigma_i on the device and blend x0 with the
        # clean latent (reference pipelines/common.py:193-232); a modality whose mask is all ones takes the uniform form
        cond = {}
        if not uniform:
            for key, st_ in (("", video_state),) + ((("audio_", audio_state),) if joint else ()):
                if not bool((st_.denoise_mask == 1).all()):
                    cond[key + "denoise_mask"] = st_.denoise_mask[0].reshape(-1).float().contiguous()
                    cond[key + "clean_latent"] = st_.clean_latent[0].float().contiguous()
        if joint:
            model.prepare(video_context, video_state.positions, per_token=not uniform, audio_context=audio_context, audio_positions=audio_state.positions)
        else:
            model.prepare(video_context, video_state.positions, per_token=not uniform)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model.capture_denoise_graph(lat, sig, audio_latent=alat, **cond)
            model.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side)
        video_state = video_state.replace(latent=lat[None].to(video_state.latent.dtype))
        if joint:
            audio_state = audio_state.replace(latent=alat[None].to(audio_state.latent.dtype))
        return video_state, audio_state
    for i in range(n):
        vm = modality_from_state(video_state, video_context, sig[i], uniform=uniform)
        if joint:
            vx0, ax0 = transformer(vm, audio_modality_from_state(audio_state, audio_context, sig[i], uniform=uniform))
        else:
            out = transformer(vm)
            vx0, ax0 = (out[0] if isinstance(out, tuple) else out), None
        if need_cfg:
            nvm = modality_from_state(video_state, negative_video_context, sig[i], uniform=uniform)
            if joint:
                nvx0, nax0 = neg(nvm, audio_modality_from_state(audio_state, negative_audio_context, sig[i], uniform=uniform))
                ax0 = audio_guider.guide(ax0, nax0) if audio_guider is not None else ax0
            else:
                nvx0 = neg(nvm)
            vx0 = video_guider.guide(vx0, nvx0) if video_guider is not None else vx0
        vx0 = post_process_latent(vx0, video_state.denoise_mask, video_state.clean_latent)
        video_state = video_state.replace(latent=stepper.step(sample=video_state.latent, denoised_sample=vx0, sigmas=sig, step_index=i))
        if joint:
            ax0 = post_process_latent(ax0, audio_state.denoise_mask, audio_state.clean_latent)
            audio_state = audio_state.replace(latent=stepper.step(sample=audio_state.latent, denoised_sample=ax0, sigmas=sig, step_index=i))
        if callback:
            callback(i + 1, n)
    return video_state, audio_state
