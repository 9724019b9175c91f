"""Distilled generation pipeline on MI355X (video path).

Mirrors reference LTX_2_MLX/pipelines/distilled.py:48-98 (DistilledConfig), :101-143
(constructor), :198-272 (_denoise_loop_av, video branch) and :274-505 (__call__): stage 1 at half
resolution with DISTILLED_SIGMA_VALUES (8 steps), optional latent 2x upscale + stage 2 with
STAGE_2_DISTILLED_SIGMA_VALUES (3 steps), then VAE decode (tiled above 4000 latent voxels).
Audio branches and image conditioning are outside this path (scope table).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Union

import torch

from ..components import (DISTILLED_SIGMA_VALUES, STAGE_2_DISTILLED_SIGMA_VALUES, EulerDiffusionStep, GaussianNoiser,
                          VideoLatentPatchifier)
from ..conditioning.tools import VideoLatentTools
from ..model.transformer import LTXModel, LTXModelType, Modality, X0Model
from ..model.upscaler import SpatialUpscaler, upscale_latent
from ..model.video_vae import SimpleVideoDecoder, TilingConfig, decode_latent, decode_tiled
from ..types import LatentState, VideoLatentShape, VideoPixelShape
from .common import modality_from_state, post_process_latent


@dataclass
class DistilledConfig:
    height: int = 480
    width: int = 704
    num_frames: int = 97          # must be 8k + 1
    seed: int = 42
    fps: float = 25.0
    tiling_config: Optional[TilingConfig] = None
    dtype: torch.dtype = torch.float32
    audio_enabled: bool = False
    use_internal_audio_branch: bool = True
    use_hip_graph: bool = False    # MI355X addition: replay the captured step loop (uniform sigma only)

    def _get_tiling_config(self) -> Optional[TilingConfig]:
        if self.tiling_config is not None:
            return self.tiling_config
        latent_frames = (self.num_frames - 1) // 8 + 1
        if latent_frames * (self.height // 32) * (self.width // 32) > 4000:
            return TilingConfig.default()
        return None

    def __post_init__(self):
        if self.num_frames % 8 != 1:
            raise ValueError(f"num_frames must be 8*k + 1, got {self.num_frames}. Valid values: 1, 9, 17, 25, 33, ..., 121")
        if self.height % 64 != 0 or self.width % 64 != 0:
            raise ValueError(f"Resolution ({self.height}x{self.width}) must be divisible by 64 for two-stage pipeline.")


class DistilledPipeline:
    def __init__(self, transformer: Union[LTXModel, X0Model], video_encoder=None, video_decoder: Optional[SimpleVideoDecoder] = None,
                 spatial_upscaler: Optional[Callable] = None, audio_decoder=None, vocoder=None):
        self.transformer = transformer if isinstance(transformer, X0Model) else X0Model(transformer)
        inner = self.transformer.velocity_model
        self.is_av_model = getattr(inner, "model_type", None) == LTXModelType.AudioVideo
        if audio_decoder is not None or vocoder is not None:
            raise NotImplementedError("audio decode is outside the MI355X hot path")
        self.video_encoder = video_encoder
        self.video_decoder = video_decoder
        self.spatial_upscaler = spatial_upscaler
        self.patchifier = VideoLatentPatchifier(patch_size=1)
        self.diffusion_step = EulerDiffusionStep()

    def _create_video_tools(self, target_shape: VideoLatentShape, fps: float) -> VideoLatentTools:
        return VideoLatentTools(patchifier=self.patchifier, target_shape=target_shape, fps=fps)

    def _denoise_loop_av(self, video_state: LatentState, audio_state, sigmas: Sequence[float], video_context: torch.Tensor,
                         audio_context=None, stepper: Optional[EulerDiffusionStep] = None,
                         callback: Optional[Callable[[int, int], None]] = None, use_hip_graph: bool = False):
        """Per step: Modality(mask*sigma) -> X0 -> post_process -> Euler.  Two equivalent
        executions: (a) API-faithful, one X0Model call + EulerDiffusionStep per step; (b) fused
        ltx2_dit_denoise_step / hipGraph replay when no callback needs intermediate states."""
        sig = [float(s) for s in sigmas]
        n = len(sig) - 1
        model = self.transformer.velocity_model
        if use_hip_graph and callback is None:
            lat = video_state.latent[0].float().contiguous()
            mask = video_state.denoise_mask.reshape(-1)
            if not bool((mask == 1).all()):
                raise ValueError("hipGraph replay needs a uniform denoise mask (no conditioning tokens)")
            model.prepare(video_context, video_state.positions)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.capture_denoise_graph(lat, sig)
                model.replay_denoise_graph()
            torch.cuda.current_stream().wait_stream(side)
            return video_state.replace(latent=lat[None].to(video_state.latent.dtype)), audio_state
        for i in range(n):
            m = modality_from_state(video_state, video_context, sig[i])
            x0 = self.transformer(m)
            x0 = post_process_latent(x0, video_state.denoise_mask, video_state.clean_latent)
            new = (stepper or self.diffusion_step).step(sample=video_state.latent, denoised_sample=x0, sigmas=sig, step_index=i)
            video_state = video_state.replace(latent=new)
            if callback:
                callback(i + 1, n)
        return video_state, audio_state

    def __call__(self, text_encoding: torch.Tensor, text_mask: Optional[torch.Tensor], config: DistilledConfig,
                 images: Optional[List] = None, callback: Optional[Callable[[str, int, int], None]] = None,
                 audio_encoding=None, initial_noise: Optional[torch.Tensor] = None):
        if images:
            raise NotImplementedError("image conditioning needs the VAE encoder (scope row f4)")
        if config.audio_enabled:
            raise NotImplementedError("audio generation is outside the MI355X hot path")
        dev = self.transformer.velocity_model.device
        gen = torch.Generator(device=dev).manual_seed(config.seed)
        noiser = GaussianNoiser(generator=gen)

        s1 = VideoPixelShape(batch=1, frames=config.num_frames, height=config.height // 2, width=config.width // 2, fps=config.fps)
        shape1 = VideoLatentShape.from_pixel_shape(s1, latent_channels=128)
        tools = self._create_video_tools(shape1, config.fps)
        state = tools.create_initial_state(dtype=config.dtype, device=dev)
        state = noiser(state, noise_scale=1.0, noise=initial_noise)
        cb1 = (lambda s, t: callback("stage1", s, t)) if callback else None
        state, _ = self._denoise_loop_av(state, None, DISTILLED_SIGMA_VALUES, text_encoding.to(dev), callback=cb1,
                                         use_hip_graph=config.use_hip_graph)
        state = tools.unpatchify(tools.clear_conditioning(state))
        final_latent = state.latent

        if self.spatial_upscaler is not None:
            # un_normalize -> upscaler -> normalize (reference pipelines/distilled.py:394-405); the statistics
            # ship with the VAE weights, so the decoder serves when no encoder object is passed
            stats = getattr(self.video_encoder, "per_channel_statistics", None) or getattr(self.video_decoder, "per_channel_statistics", None)
            if stats is None:
                raise ValueError("spatial_upscaler needs per_channel_statistics (un_normalize/normalize) from the video VAE")
            if isinstance(self.spatial_upscaler, SpatialUpscaler):
                up = upscale_latent(final_latent, self.spatial_upscaler, stats.mean_of_means, stats.std_of_means)
            else:
                up = stats.normalize(self.spatial_upscaler(stats.un_normalize(final_latent)))
            s2 = VideoPixelShape(batch=1, frames=config.num_frames, height=config.height, width=config.width, fps=config.fps)
            tools2 = self._create_video_tools(VideoLatentShape.from_pixel_shape(s2, latent_channels=128), config.fps)
            state2 = tools2.create_initial_state(dtype=config.dtype, initial_latent=up)
            state2 = noiser(state2, noise_scale=float(STAGE_2_DISTILLED_SIGMA_VALUES[0]))
            cb2 = (lambda s, t: callback("stage2", s, t)) if callback else None
            state2, _ = self._denoise_loop_av(state2, None, STAGE_2_DISTILLED_SIGMA_VALUES, text_encoding.to(dev), callback=cb2,
                                              use_hip_graph=config.use_hip_graph)
            final_latent = tools2.unpatchify(tools2.clear_conditioning(state2)).latent

        if self.video_decoder is None:
            return final_latent
        tiling = config._get_tiling_config()
        if tiling:
            chunks = list(decode_tiled(final_latent, self.video_decoder, tiling))
            return torch.cat(chunks, dim=2) if len(chunks) > 1 else chunks[0]
        return decode_latent(final_latent, self.video_decoder)


def create_distilled_pipeline(transformer, video_encoder, video_decoder, spatial_upscaler=None, audio_decoder=None, vocoder=None):
    return DistilledPipeline(transformer, video_encoder, video_decoder, spatial_upscaler, audio_decoder, vocoder)
