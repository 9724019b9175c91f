"""Checkpoint loading for the hot path (mirrors LTX_2_MLX/loader/__init__.py exports used by it)."""
from .weight_converter import (convert_pytorch_key, is_fp8_checkpoint, load_av_transformer_weights,  # noqa: F401
                               load_transformer_weights)
from .lora_loader import LoRAConfig, compute_lora_delta, find_lora_keys_for_weight, fuse_lora_into_weights, load_lora_weights  # noqa: F401
from ..model.video_vae import load_vae_decoder_weights  # noqa: F401
from ..model.video_vae_encoder import load_vae_encoder_weights  # noqa: F401
from ..model.upscaler import load_spatial_upscaler_weights  # noqa: F401
