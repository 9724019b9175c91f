"""Latent-index conditioning (image-to-video): replace the tokens of one latent frame range with an encoded
latent and set their denoise strength.  Mirrors reference LTX_2_MLX/conditioning/latent.py:9-117."""
from __future__ import annotations

import torch

from ..types import LatentState
from .tools import VideoLatentTools


class ConditioningError(Exception):
    """Raised when a conditioning item cannot be applied."""


class VideoConditionByLatentIndex:
    def __init__(self, latent: torch.Tensor, strength: float, latent_idx: int):
        """latent: (B, C, F_cond, H, W); strength: 0 keeps the tokens clean, 1 fully denoises them."""
        self.latent, self.strength, self.latent_idx = latent, strength, latent_idx

    def apply_to(self, latent_state: LatentState, latent_tools: VideoLatentTools) -> LatentState:
        cb, cc, _, ch, cw = self.latent.shape
        tgt = latent_tools.target_shape
        if (cb, cc, ch, cw) != (tgt.batch, tgt.channels, tgt.height, tgt.width):
            raise ConditioningError(f"Cannot apply image conditioning item to latent with shape {tgt}. Expected shape is "
                                    f"({tgt.batch}, {tgt.channels}, _, {tgt.height}, {tgt.width}). "
                                    "Make sure the image and latent have the same spatial shape.")
        tokens = latent_tools.patchifier.patchify(self.latent).to(latent_state.latent.device, latent_state.latent.dtype)
        start = latent_tools.patchifier.get_token_count(tgt._replace(frames=self.latent_idx))
        stop = start + tokens.shape[1]
        max_tokens = latent_tools.patchifier.get_token_count(tgt)
        if stop > max_tokens:
            raise ValueError(f"Conditioning tokens exceed latent sequence length: stop_token={stop} > max_tokens={max_tokens}. "
                             f"latent_idx={self.latent_idx}, tokens.shape={tuple(tokens.shape)}")
        mask = torch.full((tokens.shape[0], tokens.shape[1], 1), 1.0 - self.strength, dtype=latent_state.denoise_mask.dtype,
                          device=latent_state.denoise_mask.device)

        def splice(t: torch.Tensor, new: torch.Tensor) -> torch.Tensor:
            return torch.cat([t[:, :start], new, t[:, stop:]], dim=1)

        return LatentState(latent=splice(latent_state.latent, tokens), denoise_mask=splice(latent_state.denoise_mask, mask),
                           positions=latent_state.positions, clean_latent=splice(latent_state.clean_latent, tokens))
