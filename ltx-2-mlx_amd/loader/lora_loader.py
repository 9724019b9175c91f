"""LoRA loading and load-time fusion W <- W + sum_i strength_i * (B_i @ A_i)
(reference LTX_2_MLX/loader/lora_loader.py:10-19 LoRAConfig, :22-49 load_lora_weights, :52-96
find_lora_keys_for_weight, :99-126 compute_lora_delta, :129-194 fuse_lora_into_weights).

The low-rank products run on the GPU through the bf16 MFMA GEMM (`ltx2_gemm_bf16`, fp32 accumulate, fp32 out);
the sum with the base weight is formed in fp32 and rounded once to the resident dtype, as the reference does.
Fusion happens on the checkpoint-keyed state dict BEFORE `LTXModel.load_state_dict` packs q/k/v together."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .. import _native as nv
from .. import kernels as K

BF16 = torch.bfloat16


@dataclass
class LoRAConfig:
    path: str
    strength: float = 1.0

    def __post_init__(self):
        if not -2.0 <= self.strength <= 2.0:
            raise ValueError(f"LoRA strength should be between -2.0 and 2.0, got {self.strength}")


def load_lora_weights(path: str, device="cuda") -> Dict[str, torch.Tensor]:
    """safetensors -> device tensors (bf16/fp8 widened to fp32 like the reference, kept on the GPU)."""
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k).to(device).float()
    return out


def find_lora_keys_for_weight(lora_weights: Dict[str, torch.Tensor], base_key: str) -> Tuple[Optional[str], Optional[str]]:
    prefix = base_key.replace(".weight", "")
    cands = [prefix]
    if not prefix.startswith("diffusion_model."):
        cands.append(f"diffusion_model.{prefix}")
    if prefix.startswith("model."):
        cands.append(prefix.replace("model.", "diffusion_model."))
    for c in cands:
        for sa, sb in ((".lora_A.weight", ".lora_B.weight"), (".lora_down.weight", ".lora_up.weight"), (".lora_A", ".lora_B"),
                       (".lora_down", ".lora_up")):
            if c + sa in lora_weights and c + sb in lora_weights:
                return c + sa, c + sb
    return None, None


def compute_lora_delta(lora_weights: Dict[str, torch.Tensor], key_a: str, key_b: str, strength: float = 1.0) -> torch.Tensor:
    """strength * (B @ A), fp32 [out, in]: A (rank, in), B (out, rank); the rank is zero-padded to the GEMM's K granule."""
    a, b = lora_weights[key_a], lora_weights[key_b]
    r = a.shape[0]
    rp = (r + 63) // 64 * 64
    at = torch.zeros(a.shape[1], rp, device=a.device, dtype=BF16)
    at[:, :r] = a.t().to(BF16)
    bp = torch.zeros(b.shape[0], rp, device=b.device, dtype=BF16)
    bp[:, :r] = b.to(BF16)
    return K.gemm(bp, at, None, epilogue=nv.EPI_F32) * strength


def fuse_lora_into_weights(model_weights: Dict[str, torch.Tensor], lora_configs: List[LoRAConfig],
                           target_dtype: Optional[torch.dtype] = None, verbose: bool = True) -> Dict[str, torch.Tensor]:
    loras = []
    for cfg in lora_configs:
        if verbose:
            print(f"Loading LoRA: {cfg.path} (strength={cfg.strength})")
        dev = next(iter(model_weights.values())).device
        loras.append((load_lora_weights(cfg.path, dev), cfg.strength))
    fused, n_fused, n_skipped = {}, 0, 0
    for key, base in model_weights.items():
        out_dtype = target_dtype or base.dtype
        acc, applied = None, False
        for lw, strength in loras:
            ka, kb = find_lora_keys_for_weight(lw, key)
            if ka is None:
                continue
            delta = compute_lora_delta(lw, ka, kb, strength)
            if tuple(delta.shape) == tuple(base.shape):
                acc = (base.float() if acc is None else acc) + delta
                applied = True
            elif verbose:
                print(f"  Shape mismatch for {key}: base={tuple(base.shape)}, delta={tuple(delta.shape)}")
        n_fused += applied
        n_skipped += not applied
        fused[key] = acc.to(out_dtype) if applied else base.to(out_dtype)
    if verbose:
        print(f"Fused LoRA into {n_fused} weights, skipped {n_skipped}")
    return fused
