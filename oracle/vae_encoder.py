"""Oracle: video VAE encoder (image / video -> normalised latent), PyTorch fp32 on CPU (test infrastructure).

Restates LTX_2_MLX/model/video_vae/simple_encoder.py: patchify 4x4 with (c, p, r_w, r_h) channel packing
(ops.py:44-60), Conv3dSimple with ZERO spatial padding and causal (first-frame-replicated) temporal padding
(:44-75), EncoderResBlock3d pixel_norm -> SiLU -> conv x2 + residual (:121-158), SpaceToDepthDownsample3d
conv -> space-to-depth + group-mean residual with the first frame duplicated when the temporal stride is 2
(:183-257), final pixel_norm + SiLU + conv_out, first 128 channels, per-channel normalisation (:386-411).
Weights keyed as the checkpoint (`vae.encoder.*`, `vae.per_channel_statistics.*`)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# (kind, arg): res = number of blocks; down = (out_channels, stride)   (simple_encoder.py:293-306)
DEFAULT_BLOCKS = [("res", 4), ("down", (256, (1, 2, 2))), ("res", 6), ("down", (512, (2, 1, 1))), ("res", 6),
                  ("down", (1024, (2, 2, 2))), ("res", 2), ("down", (1024, (2, 2, 2))), ("res", 2)]


def patchify(x: Tensor, q: int = 4) -> Tensor:
    b, c, f, h, w = x.shape
    x = x.reshape(b, c, f, 1, h // q, q, w // q, q).permute(0, 1, 3, 7, 5, 2, 4, 6)
    return x.reshape(b, c * q * q, f, h // q, w // q)


def pixel_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    return x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + eps)


def conv3d_causal(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    x = F.pad(x, (1, 1, 1, 1, 0, 0))
    x = torch.cat([x[:, :, :1].repeat(1, 1, 2, 1, 1), x], dim=2)
    return F.conv3d(x, w.float(), b.float())


def space_to_depth(x: Tensor, stride: Tuple[int, int, int]) -> Tensor:
    b, c, t, h, w = x.shape
    st, sh, sw = stride
    x = x.reshape(b, c, t // st, st, h // sh, sh, w // sw, sw).permute(0, 1, 3, 5, 7, 2, 4, 6)
    return x.reshape(b, c * st * sh * sw, t // st, h // sh, w // sw)


def downsample(x: Tensor, w: Dict[str, Tensor], p: str, out_channels: int, stride: Tuple[int, int, int]) -> Tensor:
    if stride[0] == 2:
        x = torch.cat([x[:, :, :1], x], dim=2)
    xin = space_to_depth(x, stride)
    b, _, t, h, wd = xin.shape
    xin = xin.reshape(b, out_channels, -1, t, h, wd).mean(dim=2)
    return space_to_depth(conv3d_causal(x, w[p + ".conv.conv.weight"], w[p + ".conv.conv.bias"]), stride) + xin


def encoder_forward(video: Tensor, w: Dict[str, Tensor], blocks=DEFAULT_BLOCKS) -> Tensor:
    """SimpleVideoEncoder.__call__ (simple_encoder.py:311-411): video (B,3,F,H,W) in [-1,1] -> latent (B,128,F',H/32,W/32)."""
    if (video.shape[2] - 1) % 8 != 0:
        raise ValueError(f"Invalid number of frames: {video.shape[2]}. Encoder input must have 1 + 8*k frames (e.g., 1, 9, 17, 25, 33...).")
    x = conv3d_causal(patchify(video.float()), w["vae.encoder.conv_in.conv.weight"], w["vae.encoder.conv_in.conv.bias"])
    for i, (kind, arg) in enumerate(blocks):
        p = f"vae.encoder.down_blocks.{i}"
        if kind == "res":
            for j in range(arg):
                r = x
                x = conv3d_causal(F.silu(pixel_norm(x)), w[f"{p}.res_blocks.{j}.conv1.conv.weight"], w[f"{p}.res_blocks.{j}.conv1.conv.bias"])
                x = conv3d_causal(F.silu(pixel_norm(x)), w[f"{p}.res_blocks.{j}.conv2.conv.weight"], w[f"{p}.res_blocks.{j}.conv2.conv.bias"])
                x = x + r
        else:
            x = downsample(x, w, p, arg[0], arg[1])
    x = conv3d_causal(F.silu(pixel_norm(x)), w["vae.encoder.conv_out.conv.weight"], w["vae.encoder.conv_out.conv.bias"])
    mean = w["vae.per_channel_statistics.mean-of-means"].float().reshape(1, -1, 1, 1, 1)
    std = w["vae.per_channel_statistics.std-of-means"].float().reshape(1, -1, 1, 1, 1)
    return (x[:, :128] - mean) / std


def encoder_weight_shapes(blocks=DEFAULT_BLOCKS, base: int = 128) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, co, ci):
        s[name + ".weight"] = (co, ci, 3, 3, 3)
        s[name + ".bias"] = (co,)

    conv("vae.encoder.conv_in.conv", base, 48)
    ch = base
    for i, (kind, arg) in enumerate(blocks):
        p = f"vae.encoder.down_blocks.{i}"
        if kind == "res":
            for j in range(arg):
                conv(f"{p}.res_blocks.{j}.conv1.conv", ch, ch)
                conv(f"{p}.res_blocks.{j}.conv2.conv", ch, ch)
        else:
            out, stride = arg
            conv(f"{p}.conv.conv", out // (stride[0] * stride[1] * stride[2]), ch)
            ch = out
    conv("vae.encoder.conv_out.conv", 129, ch)
    s["vae.per_channel_statistics.mean-of-means"] = (128,)
    s["vae.per_channel_statistics.std-of-means"] = (128,)
    return s


def make_encoder_weights(seed: int = 0, blocks=DEFAULT_BLOCKS, base: int = 128) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}
    for name, shp in encoder_weight_shapes(blocks, base).items():
        if name.endswith("mean-of-means"):
            w[name] = 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("std-of-means"):
            w[name] = 0.5 + torch.rand(shp, generator=g)
        elif name.endswith(".bias"):
            w[name] = 0.02 * torch.randn(shp, generator=g)
        else:
            w[name] = torch.randn(shp, generator=g) / (27 * shp[1]) ** 0.5
    return w
