"""Checkpoint loading for the hot path (mirrors LTX_2_MLX/loader/__init__.py exports used by it)."""
from .weight_converter import (convert_pytorch_key, is_fp8_checkpoint, load_av_transformer_weights,  # noqa: F401
                               load_transformer_weights)
from ..model.video_vae import load_vae_decoder_weights  # noqa: F401
