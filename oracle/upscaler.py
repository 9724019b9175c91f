"""Oracle: spatial x2 latent upscaler, PyTorch fp32 on CPU (test infrastructure, see oracle/__init__.py).

Restates LTX_2_MLX/model/upscaler/spatial.py: conv3d with ZERO padding in all dims (:20-87), GroupNorm over
(C/groups, T, H, W) with biased variance and eps 1e-5 (:89-128), ResBlock3d conv->norm->SiLU->conv->norm->
SiLU(x + residual) (:158-181), per-frame conv2d + PixelShuffle(2) with PyTorch (C, r_h, r_w) packing and a
no-op blur at stride 1 (:184-323), SpatialUpscaler (:377-411); un_normalize / normalize bracket of the call
site (pipelines/distilled.py:394-405, video_vae/ops.py:158-186).  Weights keyed as the checkpoint."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def conv3d_zero(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return F.conv3d(x, w.float(), b.float(), padding=1)


def group_norm_5d(x: Tensor, groups: int, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    return F.group_norm(x, groups, w.float(), b.float(), eps)


def res_block(x: Tensor, w: Dict[str, Tensor], p: str, groups: int) -> Tensor:
    h = F.silu(group_norm_5d(conv3d_zero(x, w[p + ".conv1.weight"], w[p + ".conv1.bias"]), groups, w[p + ".norm1.weight"], w[p + ".norm1.bias"]))
    h = group_norm_5d(conv3d_zero(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"]), groups, w[p + ".norm2.weight"], w[p + ".norm2.bias"])
    return F.silu(h + x)


def resampler(x: Tensor, w: Dict[str, Tensor]) -> Tensor:
    b, c, f, h, wd = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, wd)
    y = F.pixel_shuffle(F.conv2d(y, w["upsampler.conv.weight"].float(), w["upsampler.conv.bias"].float(), padding=1), 2)
    return y.reshape(b, f, c, 2 * h, 2 * wd).permute(0, 2, 1, 3, 4)


def spatial_upscaler(x: Tensor, w: Dict[str, Tensor], num_blocks: int = 4, groups: int = 32) -> Tensor:
    x = F.silu(group_norm_5d(conv3d_zero(x.float(), w["initial_conv.weight"], w["initial_conv.bias"]), groups,
                             w["initial_norm.weight"], w["initial_norm.bias"]))
    for i in range(num_blocks):
        x = res_block(x, w, f"res_blocks.{i}", groups)
    x = resampler(x, w)
    for i in range(num_blocks):
        x = res_block(x, w, f"post_upsample_res_blocks.{i}", groups)
    return conv3d_zero(x, w["final_conv.weight"], w["final_conv.bias"])


def upscale_latent(latent: Tensor, w: Dict[str, Tensor], mean: Tensor, std: Tensor, num_blocks: int = 4, groups: int = 32) -> Tensor:
    m, s = mean.reshape(1, -1, 1, 1, 1), std.reshape(1, -1, 1, 1, 1)
    return (spatial_upscaler(latent.float() * s + m, w, num_blocks, groups) - m) / s


def make_upscaler_weights(in_channels: int, mid_channels: int, num_blocks: int, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def conv(name, co, ci, nd):
        shp = (co, ci) + (3,) * nd
        w[name + ".weight"] = torch.randn(shp, generator=g) / (ci * 3 ** nd) ** 0.5
        w[name + ".bias"] = 0.02 * torch.randn(co, generator=g)

    def norm(name, c):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        w[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    conv("initial_conv", mid_channels, in_channels, 3)
    norm("initial_norm", mid_channels)
    for stage in ("res_blocks", "post_upsample_res_blocks"):
        for i in range(num_blocks):
            conv(f"{stage}.{i}.conv1", mid_channels, mid_channels, 3)
            norm(f"{stage}.{i}.norm1", mid_channels)
            conv(f"{stage}.{i}.conv2", mid_channels, mid_channels, 3)
            norm(f"{stage}.{i}.norm2", mid_channels)
    conv("upsampler.conv", 4 * mid_channels, mid_channels, 2)
    conv("final_conv", in_channels, mid_channels, 3)
    return w
