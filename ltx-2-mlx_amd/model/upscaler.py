"""Spatial x2 latent upscaler on MI355X behind the reference's API
(LTX_2_MLX/model/upscaler/spatial.py:131-181 ResBlock3d, :267-323 SpatialRationalResampler, :326-411
SpatialUpscaler, :414-538 load_spatial_upscaler_weights; call site pipelines/distilled.py:394-405).

conv3d (zero padding) and the per-frame conv2d + PixelShuffle run on the implicit-GEMM MFMA kernels
(`ltx2_conv3d_fused` with pad_zero / kt=1 / depth-to-space epilogue); GroupNorm(32) + affine (+ residual)
+ SiLU is one statistics pass and one apply pass (`ltx2_groupnorm_silu`).  Activations are channels-last
bf16 in HBM; the un-normalize / normalize bracket of the call site is fused into the layout changes at
both ends (`upscale_latent`)."""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch

from .. import kernels as K

BF16 = torch.bfloat16


class _Res:
    def __init__(self):
        self.w1 = self.b1 = self.w2 = self.b2 = self.g1 = self.be1 = self.g2 = self.be2 = None


class SpatialUpscaler:
    """latent (B, 128, F, H, W) -> (B, 128, F, 2H, 2W).  Constructor keywords as reference spatial.py:343-349."""

    def __init__(self, in_channels: int = 128, mid_channels: int = 1024, num_blocks_per_stage: int = 4, num_groups: int = 32,
                 device: Union[str, torch.device] = "cuda"):
        self.in_channels, self.mid_channels, self.num_groups = in_channels, mid_channels, num_groups
        self.num_blocks_per_stage = num_blocks_per_stage
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SpatialUpscaler runs on the MI355X only (no CPU fallback); got device " + str(device))
        self._w: Dict[str, torch.Tensor] = {}
        self._loaded = False

    # ------------------------------------------------------------------ weights
    def expected_weight_shapes(self) -> Dict[str, tuple]:
        c, m = self.in_channels, self.mid_channels
        s = {"initial_conv.weight": (m, c, 3, 3, 3), "initial_conv.bias": (m,), "initial_norm.weight": (m,), "initial_norm.bias": (m,),
             "upsampler.conv.weight": (4 * m, m, 3, 3), "upsampler.conv.bias": (4 * m,),
             "final_conv.weight": (c, m, 3, 3, 3), "final_conv.bias": (c,)}
        for stage in ("res_blocks", "post_upsample_res_blocks"):
            for i in range(self.num_blocks_per_stage):
                for cv in ("conv1", "conv2"):
                    s[f"{stage}.{i}.{cv}.weight"] = (m, m, 3, 3, 3)
                    s[f"{stage}.{i}.{cv}.bias"] = (m,)
                for nm in ("norm1", "norm2"):
                    s[f"{stage}.{i}.{nm}.weight"] = (m,)
                    s[f"{stage}.{i}.{nm}.bias"] = (m,)
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """Checkpoint keys of ltx-2-spatial-upscaler-x2 (v1.0 `upsampler.conv.*` or v1.1 `upsampler.0.*`,
        spatial.py:520-538; `upsampler.blur_down.kernel` is unused at stride 1, spatial.py:234-260)."""
        sd = dict(sd)
        for a, b in (("upsampler.0.weight", "upsampler.conv.weight"), ("upsampler.0.bias", "upsampler.conv.bias")):
            if a in sd:
                sd[b] = sd.pop(a)
        exp = self.expected_weight_shapes()
        missing = [k for k in exp if k not in sd]
        if missing and strict:
            raise KeyError(f"missing {len(missing)} upscaler weights, e.g. {missing[:4]}")
        dev = self.device
        for k, shp in exp.items():
            if k not in sd:
                continue
            t = sd[k]
            if tuple(t.shape) != shp:
                raise ValueError(f"weight {k}: shape {tuple(t.shape)} != expected {shp}")
            if k == "upsampler.conv.weight":
                self._w[k] = K.conv2d_weight_to_engine(t.to(dev), pixel_shuffle=2)
            elif k == "upsampler.conv.bias":
                self._w[k] = K.conv_bias_to_engine(t.to(dev), (1, 2, 2))
            elif t.dim() == 5:
                self._w[k] = K.conv_weight_to_engine(t.to(dev))
            else:
                self._w[k] = t.to(dev, torch.float32).contiguous()
        self._loaded = True

    def init_random_weights(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shp in self.expected_weight_shapes().items():
            if "norm" in k:
                sd[k] = (1.0 if k.endswith("weight") else 0.0) + 0.1 * torch.randn(shp, generator=g)
            elif k.endswith(".bias"):
                sd[k] = 0.02 * torch.randn(shp, generator=g)
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                sd[k] = torch.randn(shp, generator=g) / fan_in ** 0.5
        self.load_state_dict(sd)

    # ------------------------------------------------------------------ forward
    def _res_block(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        """conv1 -> norm1 -> SiLU -> conv2 -> norm2 -> SiLU(x + residual)  (spatial.py:158-181)."""
        w, G = self._w, self.num_groups
        h = K.conv3d(x, w[prefix + ".conv1.weight"], w[prefix + ".conv1.bias"], pad_zero=True)
        h = K.groupnorm_silu(h, w[prefix + ".norm1.weight"], w[prefix + ".norm1.bias"], G)
        h = K.conv3d(h, w[prefix + ".conv2.weight"], w[prefix + ".conv2.bias"], pad_zero=True)
        return K.groupnorm_silu(h, w[prefix + ".norm2.weight"], w[prefix + ".norm2.bias"], G, res=x)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """x bf16 [F,H,W,C_in] channels-last -> bf16 [F,2H,2W,C_in]."""
        if not self._loaded:
            raise RuntimeError("SpatialUpscaler: weights not loaded")
        w, G = self._w, self.num_groups
        x = K.conv3d(x, w["initial_conv.weight"], w["initial_conv.bias"], pad_zero=True)
        x = K.groupnorm_silu(x, w["initial_norm.weight"], w["initial_norm.bias"], G)
        for i in range(self.num_blocks_per_stage):
            x = self._res_block(x, f"res_blocks.{i}")
        # SpatialRationalResampler: per-frame conv2d (C -> 4C) + PixelShuffle(2); blur is a no-op at stride 1
        x = K.conv3d(x, w["upsampler.conv.weight"], w["upsampler.conv.bias"], mode=2, stride=(1, 2, 2), pad_zero=True)
        for i in range(self.num_blocks_per_stage):
            x = self._res_block(x, f"post_upsample_res_blocks.{i}")
        return K.conv3d(x, w["final_conv.weight"], w["final_conv.bias"], pad_zero=True)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, C, F, H, W) float -> (B, C, F, 2H, 2W) fp32 (reference spatial.py:377-411)."""
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (1, {self.in_channels}, F, H, W), got {tuple(x.shape)}")
        dev = self.device
        one, zero = torch.ones(self.in_channels, device=dev), torch.zeros(self.in_channels, device=dev)
        y = self.forward_nhwc(K.latent_unnormalize_nhwc(x[0].to(dev, torch.float32), zero, one))
        return K.latent_normalize_nchw(y, zero, one)[None]


def upscale_latent(latent: torch.Tensor, upscaler: SpatialUpscaler, mean_of_means: torch.Tensor,
                   std_of_means: torch.Tensor) -> torch.Tensor:
    """un_normalize -> SpatialUpscaler -> normalize (pipelines/distilled.py:394-405), the two per-channel affine
    maps fused into the NCFHW<->channels-last layout changes."""
    if latent.dim() != 5 or latent.shape[0] != 1:
        raise ValueError(f"expected (1, C, F, H, W), got {tuple(latent.shape)}")
    dev = upscaler.device
    x = K.latent_unnormalize_nhwc(latent[0].to(dev, torch.float32), mean_of_means.to(dev), std_of_means.to(dev))
    return K.latent_normalize_nchw(upscaler.forward_nhwc(x), mean_of_means.to(dev), std_of_means.to(dev))[None]


def load_spatial_upscaler_weights(upscaler: SpatialUpscaler, weights_path: str) -> None:
    """safetensors -> SpatialUpscaler (reference spatial.py:414-476)."""
    from safetensors import safe_open
    sd = {}
    with safe_open(weights_path, framework="pt") as f:
        for k in f.keys():
            sd[k] = f.get_tensor(k)
    upscaler.load_state_dict(sd, strict=True)
