"""Oracle: CausalVideoVAE decode path (SimpleVideoDecoder), PyTorch fp32 on CPU.

Test infrastructure (see oracle/__init__.py).  Citations are ``path:line``
under /root/reference.  Layout follows the reference: (B, C, T, H, W).

Weights: flat dict keyed by the checkpoint names the reference's
load_vae_decoder_weights consumes (simple_decoder.py:592-671), e.g.
``vae.decoder.up_blocks.0.res_blocks.1.conv1.conv.weight`` with conv weights
in PyTorch layout (Cout, Cin, kT, kH, kW).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# simple_decoder.py:346-361
STRIDE_MAP = {"compress_all": (2, 2, 2), "compress_time": (2, 1, 1), "compress_space": (1, 2, 2)}
DEFAULT_DECODER_BLOCKS = [
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
]


@dataclass
class VAEConfig:
    decoder_blocks: Optional[list] = None
    base_channels: int = 128
    timestep_conditioning: bool = True
    latent_channels: int = 128
    decode_noise_scale: float = 0.025   # simple_decoder.py:391

    def blocks(self) -> list:
        return self.decoder_blocks if self.decoder_blocks is not None else DEFAULT_DECODER_BLOCKS

    def plan(self) -> List[Tuple[str, dict, int]]:
        """(kind, params, in_channels) per up_block, built from reversed(decoder_blocks)
        exactly as SimpleVideoDecoder.__init__ does (simple_decoder.py:393-430)."""
        ch = self.base_channels * 8
        out = []
        for name, params in reversed(self.blocks()):
            p = {"num_layers": params} if isinstance(params, int) else dict(params)
            if name == "res_x":
                out.append(("res", p, ch))
            elif name in STRIDE_MAP:
                q = {"stride": STRIDE_MAP[name], "multiplier": p.get("multiplier", 1),
                     "residual": p.get("residual", False)}
                out.append(("upsample", q, ch))
                ch = ch // q["multiplier"]
            else:
                raise ValueError(f"Unknown decoder block: {name}")
        return out

    def final_channels(self) -> int:
        ch = self.base_channels * 8
        for kind, p, _ in self.plan():
            if kind == "upsample":
                ch //= p["multiplier"]
        return ch


def vae_timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """simple_decoder.get_timestep_embedding (:12-39): freqs exp(-ln(1e4)*i/half); [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float().reshape(-1)[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def timestep_embedder(e: Tensor, w: Dict[str, Tensor], prefix: str) -> Tensor:
    """TimestepEmbedder (:42-59): linear_1 -> SiLU -> linear_2."""
    h = F.linear(e, w[prefix + ".linear_1.weight"].float(), w[prefix + ".linear_1.bias"].float())
    h = F.silu(h)
    return F.linear(h, w[prefix + ".linear_2.weight"].float(), w[prefix + ".linear_2.bias"].float())


def conv3d_simple(x: Tensor, weight: Tensor, bias: Tensor, causal: bool = False) -> Tensor:
    """Conv3dSimple.__call__ (:90-180): reflect-pad H/W by 1, replicate-pad T
    (2 front if causal else 1 front + 1 back), 3x3x3 stride-1 conv, + bias."""
    k = weight.shape[2]
    p = (k - 1) // 2
    if p > 0:
        x = torch.cat([x[:, :, :, 1:p + 1].flip(3), x, x[:, :, :, -(p + 1):-1].flip(3)], dim=3)   # :111-113
        x = torch.cat([x[..., 1:p + 1].flip(4), x, x[..., -(p + 1):-1].flip(4)], dim=4)           # :115-117
    tp = k - 1
    if causal and tp > 0:
        x = torch.cat([x[:, :, :1].repeat(1, 1, tp, 1, 1), x], dim=2)                               # :121-124
    elif tp > 0:
        pb = tp // 2
        pa = tp - pb
        x = torch.cat([x[:, :, :1].repeat(1, 1, pb, 1, 1), x, x[:, :, -1:].repeat(1, 1, pa, 1, 1)], dim=2)  # :125-134
    return F.conv3d(x, weight.float(), bias.float())


def pixel_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    """_pixel_norm (:339-342): x * rsqrt(mean(x^2 over channels) + eps)."""
    return x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + eps)


def res_block(x: Tensor, w: Dict[str, Tensor], prefix: str, time_emb: Optional[Tensor], causal: bool) -> Tensor:
    """ResBlock3d.__call__ (:194-240). scale_shift_table rows: shift1, scale1, shift2, scale2."""
    c = x.shape[1]
    table = w[prefix + ".scale_shift_table"].float()
    if time_emb is not None:
        ss = table[None] + time_emb.reshape(time_emb.shape[0], 4, c)
    else:
        ss = table[None]
    bc = lambda v: v[:, :, None, None, None]
    shift1, scale1, shift2, scale2 = bc(ss[:, 0]), 1 + bc(ss[:, 1]), bc(ss[:, 2]), 1 + bc(ss[:, 3])
    h = F.silu(pixel_norm(x) * scale1 + shift1)
    h = conv3d_simple(h, w[prefix + ".conv1.conv.weight"], w[prefix + ".conv1.conv.bias"], causal)
    h = F.silu(pixel_norm(h) * scale2 + shift2)
    h = conv3d_simple(h, w[prefix + ".conv2.conv.weight"], w[prefix + ".conv2.conv.bias"], causal)
    return h + x


def depth_to_space(x: Tensor, c_out: int, stride: Sequence[int]) -> Tensor:
    """DepthToSpaceUpsample3d._depth_to_space (:274-285): channel = ((c*ft + a)*fh + b)*fw + d."""
    b, c, t, h, w = x.shape
    ft, fh, fw = stride
    x = x.reshape(b, c_out, ft, fh, fw, t, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4)
    return x.reshape(b, c_out, t * ft, h * fh, w * fw)


def upsample_block(x: Tensor, w: Dict[str, Tensor], prefix: str, stride, multiplier: int,
                   residual: bool, causal: bool) -> Tensor:
    """DepthToSpaceUpsample3d.__call__ (:287-313)."""
    ft, fh, fw = stride
    sp = ft * fh * fw
    cin = x.shape[1]
    res = None
    if residual:
        res = depth_to_space(x, cin // sp, stride)
        if ft > 1:
            res = res[:, :, 1:]
        res = res.repeat(1, sp // multiplier, 1, 1, 1)            # mx.tile along channels (:299-300)
    y = conv3d_simple(x, w[prefix + ".conv.conv.weight"], w[prefix + ".conv.conv.bias"], causal)
    y = depth_to_space(y, cin // multiplier, stride)
    if ft > 1:
        y = y[:, :, 1:]
    return y + res if residual else y


def unpatchify(x: Tensor, r: int = 4, p: int = 1) -> Tensor:
    """ops.unpatchify 5-D branch (ops.py:109-125): packing (c, p, r_w, r_h)."""
    b, cp, f, h, w = x.shape
    c = cp // (p * r * r)
    x = x.reshape(b, c, p, r, r, f, h, w).permute(0, 1, 5, 2, 6, 4, 7, 3)
    return x.reshape(b, c, f * p, h * r, w * r)


def decoder_forward(latent: Tensor, w: Dict[str, Tensor], cfg: VAEConfig,
                    timestep: Optional[float] = 0.05, causal: bool = False,
                    noise: Optional[Tensor] = None) -> Tensor:
    """SimpleVideoDecoder.__call__ (:446-563).

    ``noise``: the N(0,1) tensor the reference draws from the MLX RNG at :497; MLX's
    Threefry stream is not reproducible here, so parity is defined on a SUPPLIED
    noise tensor (None -> zeros, i.e. only the (1 - 0.025) scaling is applied)."""
    b = latent.shape[0]
    x = latent.float()
    scaled_t = None
    if cfg.timestep_conditioning and timestep is not None:
        mult = w.get("vae.decoder.timestep_scale_multiplier", torch.tensor(1000.0)).float()
        scaled_t = torch.full((b,), float(timestep)) * mult        # :480-483
    x = x * w["vae.per_channel_statistics.std-of-means"].float()[None, :, None, None, None]
    x = x + w["vae.per_channel_statistics.mean-of-means"].float()[None, :, None, None, None]
    if cfg.timestep_conditioning and timestep is not None:
        n = torch.zeros_like(x) if noise is None else noise.float()
        x = n * cfg.decode_noise_scale + (1.0 - cfg.decode_noise_scale) * x      # :496-498
    x = conv3d_simple(x, w["vae.decoder.conv_in.conv.weight"], w["vae.decoder.conv_in.conv.bias"], causal)
    for i, (kind, p, ch) in enumerate(cfg.plan()):
        pre = f"vae.decoder.up_blocks.{i}"
        if kind == "res":
            te = None
            if scaled_t is not None and (pre + ".time_embedder.timestep_embedder.linear_1.weight") in w:
                te = timestep_embedder(vae_timestep_embedding(scaled_t), w, pre + ".time_embedder.timestep_embedder")
            for j in range(p["num_layers"]):
                x = res_block(x, w, f"{pre}.res_blocks.{j}", te, causal)            # :325-336
        else:
            x = upsample_block(x, w, pre, p["stride"], p["multiplier"], p["residual"], causal)
    x = pixel_norm(x)
    table = w["vae.decoder.last_scale_shift_table"].float()
    cfin = cfg.final_channels()
    lte = "vae.decoder.last_time_embedder.timestep_embedder"
    if scaled_t is not None and (lte + ".linear_1.weight") in w:
        te = timestep_embedder(vae_timestep_embedding(scaled_t), w, lte).reshape(b, 2, cfin)
        ss = table[None] + te
    else:
        ss = table[None]
    shift = ss[:, 0][:, :, None, None, None]
    scale = 1 + ss[:, 1][:, :, None, None, None]
    x = F.silu(x * scale + shift)                                                   # :541-542
    x = conv3d_simple(x, w["vae.decoder.conv_out.conv.weight"], w["vae.decoder.conv_out.conv.bias"], causal)
    return unpatchify(x, 4, 1)


def latent_t_to_pixel_t(lt: int) -> int:
    pt = lt
    for _ in range(3):
        pt = pt * 2 - 1
    return pt


def temporal_chunks(T: int, chunk: int = 7, overlap: int = 2) -> List[Tuple[int, int]]:
    """The chunk walk of decode_latent (simple_decoder.py:728-747)."""
    stride = chunk - overlap
    out = []
    t = 0
    while t < T:
        end = min(t + chunk, T)
        if end - t < overlap + 1 and t > 0:
            t = max(0, end - chunk)
            end = min(t + chunk, T)
        out.append((t, end))
        if end >= T:
            break
        t += stride
    return out


def blend_chunks(chunks: List[Tensor], T: int, overlap: int = 2) -> Tensor:
    """Overlap cross-fade + trim of decode_latent (simple_decoder.py:749-790)."""
    total = latent_t_to_pixel_t(T)
    if len(chunks) == 1:
        return chunks[0][:, :, :total]
    ov_ref = latent_t_to_pixel_t(overlap)
    video = chunks[0]
    for cur in chunks[1:]:
        ov = min(ov_ref, cur.shape[2], video.shape[2])
        if ov <= 1:
            video = torch.cat([video, cur], dim=2)
            continue
        ramp = torch.linspace(0.0, 1.0, ov).reshape(1, 1, ov, 1, 1)
        blended = video[:, :, -ov:] * (1.0 - ramp) + cur[:, :, :ov] * ramp
        video = torch.cat([video[:, :, :-ov], blended, cur[:, :, ov:]], dim=2)
    return video[:, :, :total]


def to_uint8_frames(video: Tensor) -> Tensor:
    """simple_decoder.py:792-800: clip((v+1)/2,0,1)*255 -> uint8 (truncation) -> (T,H,W,3)."""
    v = torch.clamp((video + 1) / 2, 0, 1) * 255
    return v.to(torch.uint8)[0].permute(1, 2, 3, 0).contiguous()


def decode_latent(latent: Tensor, w: Dict[str, Tensor], cfg: VAEConfig, timestep: Optional[float] = 0.05,
                  temporal_chunk_size: int = 7, temporal_overlap: int = 2,
                  noise: Optional[Tensor] = None, return_float: bool = False) -> Tensor:
    """decode_latent (simple_decoder.py:676-800)."""
    if latent.ndim == 4:
        latent = latent[None]
    T = latent.shape[2]
    sl = lambda a, s, e: None if a is None else a[:, :, s:e]
    if T <= temporal_chunk_size:
        video = decoder_forward(latent, w, cfg, timestep, noise=noise)
    else:
        outs = [decoder_forward(latent[:, :, s:e], w, cfg, timestep, noise=sl(noise, s, e))
                for s, e in temporal_chunks(T, temporal_chunk_size, temporal_overlap)]
        video = blend_chunks(outs, T, temporal_overlap)
    return video if return_float else to_uint8_frames(video)


# ---------------------------------------------------------------------------
# Tiled decode  (video_vae/tiling.py:9-52,154-249,349-412)
# ---------------------------------------------------------------------------
def trapezoid_mask_1d(length: int, ramp_left: int, ramp_right: int, left_starts_from_0: bool = False) -> Tensor:
    """compute_trapezoidal_mask_1d (tiling.py:9-52)."""
    if length <= 0:
        raise ValueError("Mask length must be positive.")
    ramp_left = max(0, min(ramp_left, length))
    ramp_right = max(0, min(ramp_right, length))
    mask = torch.ones(length)
    if ramp_left > 0:
        n = ramp_left + 1 if left_starts_from_0 else ramp_left + 2
        fade = torch.linspace(0.0, 1.0, n)[:-1]
        if not left_starts_from_0:
            fade = fade[1:]
        mask = torch.cat([fade, mask[ramp_left:]])
    if ramp_right > 0:
        fade = torch.linspace(1.0, 0.0, ramp_right + 2)[1:-1]
        mask = torch.cat([mask[:-ramp_right], fade])
    return mask.clamp(0, 1)


def tiles_1d(length: int, tile: int, overlap: int) -> List[Tuple[int, int, int, int]]:
    """gen_tiles_1d (tiling.py:197-219): (start, end, ramp_left, ramp_right)."""
    if length <= tile:
        return [(0, length, 0, 0)]
    out = []
    stride = tile - overlap
    pos = 0
    while pos < length:
        end = min(pos + tile, length)
        start = max(0, end - tile)
        out.append((start, end, overlap if start > 0 else 0, overlap if end < length else 0))
        if end >= length:
            break
        pos += stride
    return out


def tile_specs(latent_shape, spatial=(512, 64), temporal=(64, 24), scale=(8, 32, 32)) -> List[dict]:
    """generate_tile_specs (tiling.py:154-249). spatial/temporal = (tile, overlap) in pixels/frames or None."""
    _, _, t, h, w = latent_shape
    st, sh, sw = scale
    th, oh = (spatial[0] // sh, spatial[1] // sh) if spatial else (h, 0)
    tw, ow = (spatial[0] // sw, spatial[1] // sw) if spatial else (w, 0)
    tt, ot = (temporal[0] // st, temporal[1] // st) if temporal else (t, 0)
    specs = []
    for (t0, t1, tl, tr) in tiles_1d(t, tt, ot):
        for (h0, h1, hl, hr) in tiles_1d(h, th, oh):
            for (w0, w1, wl, wr) in tiles_1d(w, tw, ow):
                specs.append(dict(
                    in_t=(t0, t1), in_h=(h0, h1), in_w=(w0, w1),
                    out_t=(t0 * st if t0 > 0 else 0, (t1 - 1) * st + 1 if t1 > 1 else 1),
                    out_h=(h0 * sh, h1 * sh), out_w=(w0 * sw, w1 * sw),
                    ramp_t=(tl * st, tr * st), ramp_h=(hl * sh, hr * sh), ramp_w=(wl * sw, wr * sw)))
    return specs


def decode_tiled(latent: Tensor, decoder_fn, spatial=(512, 64), temporal=(64, 24)) -> Tensor:
    """decode_tiled, live second loop (tiling.py:349-412). decoder_fn(latent_tile) -> (B,3,T,H,W)."""
    b, c, t, h, w = latent.shape
    out_t, out_h, out_w = (t - 1) * 8 + 1, h * 32, w * 32
    output = torch.zeros(b, 3, out_t, out_h, out_w)
    weights = torch.zeros(1, 1, out_t, out_h, out_w)
    for s in tile_specs(latent.shape, spatial, temporal):
        tile = decoder_fn(latent[:, :, s["in_t"][0]:s["in_t"][1], s["in_h"][0]:s["in_h"][1], s["in_w"][0]:s["in_w"][1]])
        _, _, dt, dh, dw = tile.shape
        nt = min(dt, s["out_t"][1] - s["out_t"][0])
        nh = min(dh, s["out_h"][1] - s["out_h"][0])
        nw = min(dw, s["out_w"][1] - s["out_w"][0])
        mt = trapezoid_mask_1d(nt, min(s["ramp_t"][0], nt), min(s["ramp_t"][1], nt), left_starts_from_0=(s["out_t"][0] == 0))
        mh = trapezoid_mask_1d(nh, min(s["ramp_h"][0], nh), min(s["ramp_h"][1], nh))
        mw = trapezoid_mask_1d(nw, min(s["ramp_w"][0], nw), min(s["ramp_w"][1], nw))
        mask = mt[None, None, :, None, None] * mh[None, None, None, :, None] * mw[None, None, None, None, :]
        ts, hs, ws = s["out_t"][0], s["out_h"][0], s["out_w"][0]
        output[:, :, ts:ts + nt, hs:hs + nh, ws:ws + nw] += tile[:, :, :nt, :nh, :nw] * mask
        weights[:, :, ts:ts + nt, hs:hs + nh, ws:ws + nw] += mask
    return output / torch.clamp(weights, min=1e-8)


# ---------------------------------------------------------------------------
# Synthetic weights
# ---------------------------------------------------------------------------
def vae_weight_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin):
        s[name + ".weight"] = (cout, cin, 3, 3, 3)
        s[name + ".bias"] = (cout,)

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    s["vae.per_channel_statistics.mean-of-means"] = (cfg.latent_channels,)
    s["vae.per_channel_statistics.std-of-means"] = (cfg.latent_channels,)
    conv("vae.decoder.conv_in.conv", cfg.base_channels * 8, cfg.latent_channels)
    for i, (kind, p, ch) in enumerate(cfg.plan()):
        pre = f"vae.decoder.up_blocks.{i}"
        if kind == "res":
            for j in range(p["num_layers"]):
                conv(f"{pre}.res_blocks.{j}.conv1.conv", ch, ch)
                conv(f"{pre}.res_blocks.{j}.conv2.conv", ch, ch)
                s[f"{pre}.res_blocks.{j}.scale_shift_table"] = (4, ch)
            if cfg.timestep_conditioning:
                lin(f"{pre}.time_embedder.timestep_embedder.linear_1", 4 * ch, 256)
                lin(f"{pre}.time_embedder.timestep_embedder.linear_2", 4 * ch, 4 * ch)
        else:
            sp = math.prod(p["stride"])
            conv(f"{pre}.conv.conv", sp * ch // p["multiplier"], ch)
    cf = cfg.final_channels()
    conv("vae.decoder.conv_out.conv", 48, cf)
    s["vae.decoder.last_scale_shift_table"] = (2, cf)
    if cfg.timestep_conditioning:
        s["vae.decoder.timestep_scale_multiplier"] = ()
        lin("vae.decoder.last_time_embedder.timestep_embedder.linear_1", 256, 256)
        lin("vae.decoder.last_time_embedder.timestep_embedder.linear_2", 2 * cf, 256)
    return s


def make_vae_weights(cfg: VAEConfig, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in vae_weight_shapes(cfg).items():
        if name.endswith("timestep_scale_multiplier"):
            t = torch.tensor(1000.0)
        elif name.endswith("std-of-means"):
            t = 1.0 + 0.1 * torch.rand(shape, generator=g)
        elif name.endswith("mean-of-means"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("scale_shift_table"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif len(shape) == 5:
            fan_in = shape[1] * 27
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        out[name] = t
    return out
