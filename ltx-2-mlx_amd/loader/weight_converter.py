"""safetensors -> HBM loader for the DiT (reference LTX_2_MLX/loader/weight_converter.py:277-446,527-553,
loader/fp8_loader.py:14-130).

Same entry points and keyword meaning as the reference.  What differs is where the bytes go: each
tensor is read once from the (memory-mapped) safetensors file and copied straight to HBM; fp8
(`float8_e4m3fn` + per-tensor `<key>.weight_scale`) weights are uploaded as raw bytes and dequantised
ON the GPU (`ltx2_dequant_fp8_e4m3fn`: f32(fp8) * scale -> bf16, the reference's arithmetic at
weight_converter.py:391-395 with bf16 instead of fp16 as the resident dtype).  The reference's
torch -> numpy -> MLX double copy (weight_converter.py:383-431) does not exist here.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import kernels as K

PREFIX = "model.diffusion_model."


def convert_pytorch_key(key: str, include_audio: bool = False) -> Optional[str]:
    """Key filter of convert_pytorch_key_to_mlx (weight_converter.py:277-315).  The engine keeps the
    checkpoint's own names (to_out.0, ff.net.0.proj, ff.net.2), so only the skip rules apply."""
    if not include_audio and ("av_ca" in key or "a2v" in key or "audio" in key.lower()):
        return None
    if "video_embeddings_connector" in key or "audio_embeddings_connector" in key:
        return None                      # text-encoder weights, loaded elsewhere in the reference
    return key


from ..model.transformer import FP8_RESIDENT_KEYS, Fp8Weight  # noqa: E402


def is_fp8_checkpoint(weights_path: str) -> bool:
    """True if any `<key>.weight_scale` entry exists (fp8_loader.py:133-150)."""
    from safetensors import safe_open
    with safe_open(weights_path, framework="pt") as f:
        return any(k.endswith(".weight_scale") for k in f.keys())


def load_transformer_weights(model, weights_path: str, strict: bool = False, use_fp8: bool = False,
                             include_audio: bool = False, streaming: bool = True, target_dtype: str = "float16",
                             lora_configs=None, fp8_resident: bool = False) -> None:
    """Load `model.diffusion_model.*` tensors into an LTXModel (weight_converter.py:318-446).

    use_fp8: dequantise fp8 weights with their `weight_scale` (ignored `input_scale`, fp8_loader.py:87-97);
    include_audio: keep audio / av_ca / a2v keys (AudioVideo model); `streaming` and `target_dtype` (reference default
    "float16") are accepted for signature compatibility (loading always streams; the resident dtype is bf16);
    fp8_resident (MI355X addition, BASELINE config 3): the video stream's attention / feed-forward projections stay
    float8_e4m3fn + scale in HBM (half the bytes) and are expanded inside the GEMM -- bit-identical to dequantising at load;
    lora_configs: LoRAConfig list fused into the checkpoint-keyed weights on the GPU before they are packed
    (reference loader/lora_loader.py:129-194)."""
    from safetensors import safe_open
    dev = model.device
    sd: Dict[str, torch.Tensor] = {}
    n_fp8 = n_res = 0
    if fp8_resident and lora_configs:
        raise NotImplementedError("LoRA fusion needs dequantised weights: drop fp8_resident")
    with safe_open(weights_path, framework="pt") as f:
        keys = list(f.keys())
        scales = {}
        if use_fp8:
            for k in keys:
                if k.endswith(".weight_scale"):
                    scales[k[:-len("_scale")]] = float(f.get_tensor(k).float().item())
        for full in keys:
            if not full.startswith(PREFIX) or full.endswith("_scale"):
                continue
            key = convert_pytorch_key(full[len(PREFIX):], include_audio=include_audio)
            if key is None:
                continue
            t = f.get_tensor(full)
            if full in scales:
                if t.dtype != torch.float8_e4m3fn:
                    raise ValueError(f"{full}: has a weight_scale but dtype {t.dtype}, expected float8_e4m3fn")
                if fp8_resident and FP8_RESIDENT_KEYS.match(key) and t.shape[0] % 256 == 0 and t.shape[1] % 128 == 0 and t.shape[1] >= 256:
                    sd[key] = Fp8Weight(t.view(torch.uint8).to(dev, non_blocking=True), scales[full])
                    n_res += 1
                else:
                    sd[key] = K.dequant_fp8(t.view(torch.uint8).to(dev, non_blocking=True), scales[full])
                n_fp8 += 1
            elif t.dtype == torch.float8_e4m3fn:          # fp8 without a scale (weight_converter.py:399-401)
                sd[key] = K.dequant_fp8(t.view(torch.uint8).to(dev, non_blocking=True), 1.0)
                n_fp8 += 1
            else:
                sd[key] = t.to(dev, non_blocking=True)
    if lora_configs:
        from .lora_loader import fuse_lora_into_weights
        sd = fuse_lora_into_weights(sd, lora_configs)
    model.load_state_dict(sd, strict=strict)
    print(f"  loaded {len(sd)} transformer tensors ({n_fp8} fp8, {n_res} of them kept fp8-resident) from {weights_path}")


def load_av_transformer_weights(model, weights_path: str, strict: bool = False, use_fp8: bool = False,
                                target_dtype: str = "float16") -> None:
    """load_transformer_weights(include_audio=True) (weight_converter.py:527-553)."""
    load_transformer_weights(model, weights_path, strict=strict, use_fp8=use_fp8, include_audio=True, target_dtype=target_dtype)
