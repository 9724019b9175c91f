"""Single-stage generation pipeline on MI355X: the Euler branches of the reference's OneStagePipeline, with or without classifier-free guidance.

Mirrors reference LTX_2_MLX/pipelines/one_stage.py:52-110 (OneStageCFGConfig, field names and defaults verbatim),
:113-160 (constructor), :224-330 / :466-568 (`_denoise_loop_cfg`, `_denoise_loop_cfg_av` with `need_cfg == False`) and
:731-1011 (`__call__`): LTX2Scheduler sigma schedule over `num_inference_steps`, optional image conditioning by latent
replacement, video-only or joint audio+video denoising on an AudioVideo transformer (LTX-2.3 always takes the joint
branch: `use_internal_audio_branch`), VAE decode (tiled above 4000 latent voxels).  This is the path the reference's CLI
takes for LTX-2.3 checkpoints and for `--generate-audio` (scripts/generate.py:1638-1735; distilled models run it with
cfg_scale = audio_cfg_scale = 1, i.e. one transformer evaluation per step).

Classifier-free guidance (round 3): `cfg_scale` / `audio_cfg_scale` != 1 evaluate the negative prompt too (a second engine context over the same
weights) and combine the two predictions with CFGGuider or, for `rescale_scale > 0`, CFGStarRescalingGuider -- per modality, as :793-807.

Outside the MI355X hot path, rejected with NotImplementedError: STG / APG guidance (`stg_scale`, `guider_override`), the Heun sampler, GE velocity correction, cross-attention scaling, the temporal upscaler, audio VAE /
vocoder decode (with `audio_enabled` the audio LATENT is returned in place of the waveform).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple, Union

import torch

from ..components import (AudioPatchifier, CFGGuider, CFGStarRescalingGuider, EulerDiffusionStep, GaussianNoiser, LTX2Scheduler,
                          VideoLatentPatchifier)
from ..conditioning.tools import AudioLatentTools, VideoLatentTools
from ..model.transformer import LTXModel, LTXModelType, X0Model
from ..model.video_vae import SimpleVideoDecoder, TilingConfig, decode_latent, decode_tiled
from ..types import AudioLatentShape, LatentState, VideoLatentShape, VideoPixelShape
from .common import ImageCondition, apply_conditionings, create_image_conditionings, joint_denoise_loop


@dataclass
class OneStageCFGConfig:
    """Configuration of the single-stage pipeline (reference pipelines/one_stage.py:52-110)."""
    height: int = 480
    width: int = 704
    num_frames: int = 97            # must be 8k + 1
    seed: int = 42
    fps: float = 24.0
    num_inference_steps: int = 30
    cfg_scale: float = 3.0          # video text guidance (1.0: none, one transformer evaluation per step)
    audio_cfg_scale: float = 7.0    # audio text guidance
    rescale_scale: float = 0.7      # > 0: CFGStarRescalingGuider, else CFGGuider (one_stage.py:796-805)
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
    use_hip_graph: bool = False     # MI355X addition: replay the captured step loop (uniform sigma, no callback)

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
        if self.height % 32 != 0 or self.width % 32 != 0:
            raise ValueError(f"Resolution ({self.height}x{self.width}) must be divisible by 32 for single-stage pipeline.")


class OneStagePipeline:
    def __init__(self, transformer: Union[LTXModel, X0Model], video_encoder=None, video_decoder: Optional[SimpleVideoDecoder] = None,
                 audio_decoder=None, vocoder=None):
        self.transformer = transformer if isinstance(transformer, X0Model) else X0Model(transformer)
        inner = self.transformer.velocity_model
        self.is_av_model = getattr(inner, "model_type", None) == LTXModelType.AudioVideo
        if audio_decoder is not None or vocoder is not None:
            raise NotImplementedError("audio VAE / vocoder decode is outside the MI355X hot path (the audio latent is returned)")
        self.video_encoder = video_encoder
        self.video_decoder = video_decoder
        self.audio_decoder = None
        self.vocoder = None
        self.patchifier = VideoLatentPatchifier(patch_size=1)
        self.audio_patchifier = AudioPatchifier(patch_size=1)
        self.diffusion_step = EulerDiffusionStep()
        self.scheduler = LTX2Scheduler()

    def _create_video_tools(self, target_shape: VideoLatentShape, fps: float) -> VideoLatentTools:
        return VideoLatentTools(patchifier=self.patchifier, target_shape=target_shape, fps=fps)

    def _create_audio_tools(self, target_shape: AudioLatentShape) -> AudioLatentTools:
        return AudioLatentTools(patchifier=self.audio_patchifier, target_shape=target_shape)

    @staticmethod
    def _require_no_guidance(stg_scale, guider_override, ge_gamma, sampler, temporal_upscaler, cross_attn_scale):
        """Everything beyond classifier-free guidance and the Euler step is not built here.  The reference enables STG / GE only for
        values > 0 (pipelines/one_stage.py:867, :301): a zero or negative value is a no-op there and passes here."""
        if guider_override is not None:
            raise NotImplementedError("guider_override (APG / custom guiders) is outside the MI355X hot path")
        if stg_scale > 0:
            raise NotImplementedError("STG guidance is outside the MI355X hot path")
        if ge_gamma > 0:
            raise NotImplementedError("GE velocity correction is outside the MI355X hot path")
        if sampler != "euler":
            raise NotImplementedError(f"sampler={sampler!r}: only the Euler step is built")
        if temporal_upscaler is not None:
            raise NotImplementedError("the temporal upscaler is outside the MI355X hot path")
        if cross_attn_scale != 1.0:
            raise NotImplementedError("cross_attn_scale != 1.0 is outside the MI355X hot path")

    def __call__(self, positive_encoding: torch.Tensor, negative_encoding: Optional[torch.Tensor], config: OneStageCFGConfig,
                 images: Optional[List[ImageCondition]] = None, callback: Optional[Callable[[int, int], None]] = None,
                 positive_audio_encoding: Optional[torch.Tensor] = None, negative_audio_encoding: Optional[torch.Tensor] = None,
                 stg_scale: float = 0.0, stg_blocks: Optional[List[int]] = None, stg_cutoff: float = 1.0, guider_override=None,
                 ge_gamma: float = 0.0, sampler: str = "euler", temporal_upscaler=None, cross_attn_scale: float = 1.0,
                 cross_attn_start_block: int = 40, *, initial_noise: Optional[torch.Tensor] = None,
                 initial_audio_noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """-> (video, audio): video uint8 frames (F, H, W, 3) (or the final latent when no decoder is set); audio = the audio
        LATENT (B, 8, T_a, 16) when config.audio_enabled (the reference returns the vocoder's waveform), else None.
        The negative encodings are evaluated when a guider is enabled (cfg_scale / audio_cfg_scale != 1); unlike the reference
        (one_stage.py:776-780, which demands negative_audio_encoding whenever the audio branch runs) they may be None when no guider is.  initial_noise /
        initial_audio_noise (keyword-only, MI355X addition): supplied N(0,1) tensors of the patchified latent shapes, so
        results can be compared with the oracle loop (MLX's RNG stream is not reproducible here)."""
        images = images or []
        internal_audio_active = self.is_av_model and (config.use_internal_audio_branch or config.audio_enabled)
        if config.audio_enabled or internal_audio_active:
            if positive_audio_encoding is None:
                raise ValueError("Audio encoding required for AudioVideo generation. Provide positive_audio_encoding and negative_audio_encoding.")
        if config.audio_enabled and not self.is_av_model:
            raise ValueError("audio_enabled needs an AudioVideo transformer")
        self._require_no_guidance(stg_scale, guider_override, ge_gamma, sampler, temporal_upscaler, cross_attn_scale)

        dev = self.transformer.velocity_model.device
        noiser = GaussianNoiser(generator=torch.Generator(device=dev).manual_seed(config.seed))
        pixel_shape = VideoPixelShape(batch=1, frames=config.num_frames, height=config.height, width=config.width, fps=config.fps)
        latent_shape = VideoLatentShape.from_pixel_shape(pixel_shape, latent_channels=128)
        video_tools = self._create_video_tools(latent_shape, config.fps)
        conditionings = create_image_conditionings(images, self.video_encoder, config.height, config.width, config.dtype)
        video_state = video_tools.create_initial_state(dtype=config.dtype, device=dev)
        video_state = apply_conditionings(video_state, conditionings, video_tools)
        sigmas = self.scheduler.execute(steps=config.num_inference_steps)        # no latent: the MAX_SHIFT_ANCHOR schedule (one_stage.py:837)
        video_state = noiser(video_state, noise_scale=1.0, noise=initial_noise)

        audio_state, audio_tools = None, None
        if internal_audio_active:
            audio_shape = AudioLatentShape.from_video_pixel_shape(
                pixel_shape, channels=config.audio_vae_channels, mel_bins=config.audio_mel_bins, sample_rate=config.audio_sample_rate,
                hop_length=config.audio_hop_length, audio_latent_downsample_factor=config.audio_downsample_factor)
            audio_tools = self._create_audio_tools(audio_shape)
            audio_state = noiser(audio_tools.create_initial_state(dtype=config.dtype, device=dev), noise_scale=1.0, noise=initial_audio_noise)

        actx = positive_audio_encoding.to(dev) if (internal_audio_active and positive_audio_encoding is not None) else None
        # one guider per modality (one_stage.py:793-807)
        mk = CFGStarRescalingGuider if config.rescale_scale > 0 else CFGGuider
        video_guider, audio_guider = mk(scale=config.cfg_scale), mk(scale=config.audio_cfg_scale)
        need_cfg = video_guider.enabled() or (internal_audio_active and audio_guider.enabled())
        if need_cfg and (negative_encoding is None or (internal_audio_active and negative_audio_encoding is None)):
            raise ValueError("cfg_scale / audio_cfg_scale != 1 need the negative prompt's encoding(s)")
        nactx = negative_audio_encoding.to(dev) if (need_cfg and internal_audio_active) else None
        video_state, audio_state = joint_denoise_loop(self.transformer, self.is_av_model, video_state, audio_state, sigmas,
                                                      positive_encoding.to(dev), actx, self.diffusion_step, callback, config.use_hip_graph,
                                                      negative_video_context=negative_encoding.to(dev) if need_cfg else None,
                                                      negative_audio_context=nactx, video_guider=video_guider, audio_guider=audio_guider)

        video_state = video_tools.unpatchify(video_tools.clear_conditioning(video_state))
        final_video_latent = video_state.latent
        if self.video_decoder is None:
            video = final_video_latent
        else:
            tiling = config._get_tiling_config()
            if tiling:
                chunks = list(decode_tiled(final_video_latent, self.video_decoder, tiling))
                video = torch.cat(chunks, dim=2) if len(chunks) > 1 else chunks[0]
            else:
                video = decode_latent(final_video_latent, self.video_decoder)
        audio = None
        if config.audio_enabled and audio_state is not None:
            audio = audio_tools.unpatchify(audio_tools.clear_conditioning(audio_state)).latent
        return video, audio


def create_one_stage_pipeline(transformer, video_encoder, video_decoder, audio_decoder=None, vocoder=None) -> OneStagePipeline:
    return OneStagePipeline(transformer, video_encoder, video_decoder, audio_decoder, vocoder)
