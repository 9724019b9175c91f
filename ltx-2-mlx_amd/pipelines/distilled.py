"""Distilled generation pipeline on MI355X (video path).

Mirrors reference LTX_2_MLX/pipelines/distilled.py:48-98 (DistilledConfig), :101-143
(constructor), :198-272 (_denoise_loop_av, video branch) and :274-505 (__call__): stage 1 at half
resolution with DISTILLED_SIGMA_VALUES (8 steps), optional latent 2x upscale + stage 2 with
STAGE_2_DISTILLED_SIGMA_VALUES (3 steps), then VAE decode (tiled above 4000 latent voxels).
The joint audio+video branch runs on AudioVideo transformers and returns the audio latent; audio VAE / vocoder
decode is outside this path (scope table).  Image conditioning encodes the image with the VAE encoder
and runs the per-token-sigma path of the DiT.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Union

import torch

from ..components import (DISTILLED_SIGMA_VALUES, STAGE_2_DISTILLED_SIGMA_VALUES, AudioPatchifier, EulerDiffusionStep,
                          GaussianNoiser, VideoLatentPatchifier)
from ..conditioning.tools import AudioLatentTools, VideoLatentTools
from ..model.transformer import LTXModel, LTXModelType, Modality, X0Model
from ..model.upscaler import SpatialUpscaler, upscale_latent
from ..model.video_vae import SimpleVideoDecoder, TilingConfig, decode_latent, decode_tiled
from ..types import AudioLatentShape, LatentState, VideoLatentShape, VideoPixelShape
from .common import apply_conditionings, create_image_conditionings, joint_denoise_loop


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
    audio_vae_channels: int = 8
    audio_mel_bins: int = 16
    audio_sample_rate: int = 16000
    audio_hop_length: int = 160
    audio_downsample_factor: int = 4
    audio_output_sample_rate: int = 24000
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
            raise NotImplementedError("audio VAE / vocoder decode is outside the MI355X hot path (the audio latent is returned)")
        self.video_encoder = video_encoder
        self.video_decoder = video_decoder
        self.spatial_upscaler = spatial_upscaler
        self.patchifier = VideoLatentPatchifier(patch_size=1)
        self.diffusion_step = EulerDiffusionStep()

    def _create_video_tools(self, target_shape: VideoLatentShape, fps: float) -> VideoLatentTools:
        return VideoLatentTools(patchifier=self.patchifier, target_shape=target_shape, fps=fps)

    def _create_audio_tools(self, target_shape: AudioLatentShape) -> AudioLatentTools:
        return AudioLatentTools(patchifier=AudioPatchifier(patch_size=1), target_shape=target_shape)

    @staticmethod
    def _channelwise_normalize_audio(latent: torch.Tensor) -> torch.Tensor:
        """Length-invariant audio noise (reference pipelines/distilled.py:165-187): global zero-mean / unit-std
        (population std), then per-feature standardisation over the token axis."""
        x = (latent - latent.mean()) / (latent.std(unbiased=False) + 1e-8)
        return (x - x.mean(dim=1, keepdim=True)) / (x.std(dim=1, keepdim=True, unbiased=False) + 1e-8)

    def _denoise_loop_av(self, video_state: LatentState, audio_state: Optional[LatentState], sigmas: Sequence[float],
                         video_context: torch.Tensor, audio_context: Optional[torch.Tensor] = None,
                         stepper: Optional[EulerDiffusionStep] = None, callback: Optional[Callable[[int, int], None]] = None,
                         use_hip_graph: bool = False):
        """Joint audio+video loop (reference pipelines/distilled.py:198-271): see pipelines.common.joint_denoise_loop."""
        return joint_denoise_loop(self.transformer, self.is_av_model, video_state, audio_state, sigmas, video_context, audio_context,
                                  stepper or self.diffusion_step, callback, use_hip_graph)

    def __call__(self, text_encoding: torch.Tensor, text_mask: Optional[torch.Tensor], config: DistilledConfig,
                 images: Optional[List] = None, callback: Optional[Callable[[str, int, int], None]] = None,
                 audio_encoding: Optional[torch.Tensor] = None, initial_noise: Optional[torch.Tensor] = None,
                 initial_audio_noise: Optional[torch.Tensor] = None, stage2_noise: Optional[torch.Tensor] = None):
        """Returns the decoded video (uint8 frames, or the final latent when no decoder is set); with
        config.audio_enabled on an AudioVideo model, the tuple (video, audio_latent) -- the audio VAE / vocoder
        that turn the (B, 8, T_a, 16) latent into a waveform are outside this path."""
        images = images or []
        dev = self.transformer.velocity_model.device
        gen = torch.Generator(device=dev).manual_seed(config.seed)
        noiser = GaussianNoiser(generator=gen)
        audio_active = self.is_av_model and (config.use_internal_audio_branch or config.audio_enabled)
        if config.audio_enabled and not self.is_av_model:
            raise ValueError("audio_enabled needs an AudioVideo transformer")
        actx = audio_encoding.to(dev) if audio_encoding is not None else None

        def audio_shape(pix: VideoPixelShape) -> AudioLatentShape:
            return AudioLatentShape.from_video_pixel_shape(pix, channels=config.audio_vae_channels, mel_bins=config.audio_mel_bins,
                                                           sample_rate=config.audio_sample_rate, hop_length=config.audio_hop_length,
                                                           audio_latent_downsample_factor=config.audio_downsample_factor)

        s1 = VideoPixelShape(batch=1, frames=config.num_frames, height=config.height // 2, width=config.width // 2, fps=config.fps)
        shape1 = VideoLatentShape.from_pixel_shape(s1, latent_channels=128)
        tools = self._create_video_tools(shape1, config.fps)
        state = tools.create_initial_state(dtype=config.dtype, device=dev)
        # image conditioning at stage-1 resolution: encoded latent replaces the tokens of its latent frame and
        # lowers their denoise mask (reference pipelines/distilled.py:326-333)
        state = apply_conditionings(state, create_image_conditionings(images, self.video_encoder, s1.height, s1.width, config.dtype), tools)
        state = noiser(state, noise_scale=1.0, noise=initial_noise)
        astate, atools = None, None
        if audio_active:
            atools = self._create_audio_tools(audio_shape(s1))
            astate = noiser(atools.create_initial_state(dtype=config.dtype, device=dev), noise_scale=1.0, noise=initial_audio_noise)
            astate = astate.replace(latent=self._channelwise_normalize_audio(astate.latent))
        cb1 = (lambda s, t: callback("stage1", s, t)) if callback else None
        state, astate = self._denoise_loop_av(state, astate, DISTILLED_SIGMA_VALUES, text_encoding.to(dev), actx, callback=cb1,
                                              use_hip_graph=config.use_hip_graph)
        state = tools.unpatchify(tools.clear_conditioning(state))
        final_latent = state.latent
        audio_latent = atools.unpatchify(atools.clear_conditioning(astate)).latent if astate is not None else None

        if self.spatial_upscaler is not None:
            # un_normalize -> upscaler -> normalize (reference pipelines/distilled.py:394-405); the statistics
            # ship with the VAE weights, so the decoder serves when no encoder object is passed
            # the encoder's statistics only when it actually carries loaded weights (an unloaded SimpleVideoEncoder holds the
            # identity placeholder zeros / ones, which would silently mis-scale stage 2); else the decoder's
            enc_stats = getattr(self.video_encoder, "per_channel_statistics", None)
            if enc_stats is not None and not getattr(self.video_encoder, "_loaded", True):
                enc_stats = None
            stats = enc_stats or getattr(self.video_decoder, "per_channel_statistics", None)
            if stats is None:
                raise ValueError("spatial_upscaler needs per_channel_statistics (un_normalize/normalize) from the video VAE")
            if isinstance(self.spatial_upscaler, SpatialUpscaler):
                up = upscale_latent(final_latent, self.spatial_upscaler, stats.mean_of_means, stats.std_of_means)
            else:
                up = stats.normalize(self.spatial_upscaler(stats.un_normalize(final_latent)))
            s2 = VideoPixelShape(batch=1, frames=config.num_frames, height=config.height, width=config.width, fps=config.fps)
            tools2 = self._create_video_tools(VideoLatentShape.from_pixel_shape(s2, latent_channels=128), config.fps)
            state2 = tools2.create_initial_state(dtype=config.dtype, initial_latent=up)
            state2 = apply_conditionings(state2, create_image_conditionings(images, self.video_encoder, s2.height, s2.width, config.dtype), tools2)
            sigma0 = float(STAGE_2_DISTILLED_SIGMA_VALUES[0])
            state2 = noiser(state2, noise_scale=sigma0, noise=stage2_noise)
            astate2, atools2 = None, None
            if audio_active:        # no spatial upscaling for audio: stage 1's latent is re-noised (reference :441-458)
                atools2 = self._create_audio_tools(audio_shape(s2))
                astate2 = noiser(atools2.create_initial_state(dtype=config.dtype, initial_latent=audio_latent), noise_scale=sigma0)
            cb2 = (lambda s, t: callback("stage2", s, t)) if callback else None
            state2, astate2 = self._denoise_loop_av(state2, astate2, STAGE_2_DISTILLED_SIGMA_VALUES, text_encoding.to(dev), actx,
                                                    callback=cb2, use_hip_graph=config.use_hip_graph)
            final_latent = tools2.unpatchify(tools2.clear_conditioning(state2)).latent
            if astate2 is not None:
                audio_latent = atools2.unpatchify(atools2.clear_conditioning(astate2)).latent

        if self.video_decoder is None:
            video = final_latent
        else:
            tiling = config._get_tiling_config()
            if tiling:
                chunks = list(decode_tiled(final_latent, self.video_decoder, tiling))
                video = torch.cat(chunks, dim=2) if len(chunks) > 1 else chunks[0]
            else:
                video = decode_latent(final_latent, self.video_decoder)
        return (video, audio_latent) if config.audio_enabled else video


def create_distilled_pipeline(transformer, video_encoder, video_decoder, spatial_upscaler=None, audio_decoder=None, vocoder=None):
    return DistilledPipeline(transformer, video_encoder, video_decoder, spatial_upscaler, audio_decoder, vocoder)
