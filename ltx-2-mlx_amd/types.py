"""Shape / state types of the sampling loop (host-side; mirrors reference LTX_2_MLX/types.py:10-194
for the video and audio latent shapes).  Tensors are torch tensors on the GPU."""
from __future__ import annotations

from dataclasses import dataclass
from typing import NamedTuple, Tuple

import torch


class VideoPixelShape(NamedTuple):
    batch: int
    frames: int
    height: int
    width: int
    fps: float = 25.0


class SpatioTemporalScaleFactors(NamedTuple):
    time: int
    width: int
    height: int

    @classmethod
    def default(cls) -> "SpatioTemporalScaleFactors":
        return cls(time=8, width=32, height=32)


VIDEO_SCALE_FACTORS = SpatioTemporalScaleFactors.default()


class VideoLatentShape(NamedTuple):
    batch: int
    channels: int
    frames: int
    height: int
    width: int

    def to_tuple(self) -> Tuple[int, int, int, int, int]:
        return (self.batch, self.channels, self.frames, self.height, self.width)

    @staticmethod
    def from_shape(shape) -> "VideoLatentShape":
        return VideoLatentShape(*[int(s) for s in shape[:5]])

    def mask_shape(self) -> "VideoLatentShape":
        return self._replace(channels=1)

    @staticmethod
    def from_pixel_shape(shape: VideoPixelShape, latent_channels: int = 128,
                         scale_factors: SpatioTemporalScaleFactors = VIDEO_SCALE_FACTORS) -> "VideoLatentShape":
        # reference types.py:72-87
        return VideoLatentShape(batch=shape.batch, channels=latent_channels,
                                frames=(shape.frames - 1) // scale_factors.time + 1,
                                height=shape.height // scale_factors.height,
                                width=shape.width // scale_factors.width)

    def upscale(self, scale_factors: SpatioTemporalScaleFactors = VIDEO_SCALE_FACTORS) -> "VideoLatentShape":
        return self._replace(channels=3, frames=(self.frames - 1) * scale_factors.time + 1,
                             height=self.height * scale_factors.height, width=self.width * scale_factors.width)


class AudioLatentShape(NamedTuple):
    """Audio latent (batch, channels, frames, mel_bins) (reference types.py:100-164)."""
    batch: int
    channels: int
    frames: int
    mel_bins: int

    def to_tuple(self) -> Tuple[int, int, int, int]:
        return (self.batch, self.channels, self.frames, self.mel_bins)

    def mask_shape(self) -> "AudioLatentShape":
        return self._replace(channels=1, mel_bins=1)

    @staticmethod
    def from_shape(shape) -> "AudioLatentShape":
        return AudioLatentShape(*[int(s) for s in shape[:4]])

    @staticmethod
    def from_duration(batch: int, duration: float, channels: int = 8, mel_bins: int = 16, sample_rate: int = 16000,
                      hop_length: int = 160, audio_latent_downsample_factor: int = 4) -> "AudioLatentShape":
        latents_per_second = float(sample_rate) / float(hop_length) / float(audio_latent_downsample_factor)
        return AudioLatentShape(batch=batch, channels=channels, frames=round(duration * latents_per_second), mel_bins=mel_bins)

    @staticmethod
    def from_video_pixel_shape(shape: VideoPixelShape, channels: int = 8, mel_bins: int = 16, sample_rate: int = 16000,
                               hop_length: int = 160, audio_latent_downsample_factor: int = 4) -> "AudioLatentShape":
        return AudioLatentShape.from_duration(batch=shape.batch, duration=float(shape.frames) / float(shape.fps), channels=channels,
                                              mel_bins=mel_bins, sample_rate=sample_rate, hop_length=hop_length,
                                              audio_latent_downsample_factor=audio_latent_downsample_factor)


@dataclass(frozen=True)
class LatentState:
    """latent / denoise_mask / positions / clean_latent (reference types.py:167-194)."""
    latent: torch.Tensor
    denoise_mask: torch.Tensor
    positions: torch.Tensor
    clean_latent: torch.Tensor

    def replace(self, **kw) -> "LatentState":
        return LatentState(latent=kw.get("latent", self.latent), denoise_mask=kw.get("denoise_mask", self.denoise_mask),
                           positions=kw.get("positions", self.positions), clean_latent=kw.get("clean_latent", self.clean_latent))
