"""CausalVideoVAE decode behind the reference's API: SimpleVideoDecoder, decode_latent,
decode_tiled, TilingConfig, load_vae_decoder_weights.

Mirrors reference LTX_2_MLX/model/video_vae/simple_decoder.py:364-563 (decoder), :566-673 (loader),
:676-800 (decode_latent) and tiling.py:9-412.  The decoder pass runs in libltx2hip.so
(ltx2_vae_decode: implicit-GEMM conv3d + fused norm/activation kernels, channels-last bf16);
torch owns memory, and does the cheap chunk/tile bookkeeping (slicing, cross-fades).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Dict, Iterator, List, Optional, Tuple, Union

import torch

from .. import _native as nv
from .. import kernels as K

BF16 = torch.bfloat16

_STRIDE_MAP = {"compress_all": (2, 2, 2), "compress_time": (2, 1, 1), "compress_space": (1, 2, 2)}
# Default V2.0 decoder blocks (reference simple_decoder.py:353-361)
_DEFAULT_DECODER_BLOCKS = [
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
    ["compress_all", {"multiplier": 2, "residual": True}],
    ["res_x", {"num_layers": 5}],
]


class SimpleVideoDecoder:
    """Config-driven decoder (architecture from checkpoint metadata `decoder_blocks`)."""

    def __init__(self, decoder_blocks: Optional[List] = None, base_channels: int = 128, timestep_conditioning: bool = True,
                 compute_dtype: torch.dtype = BF16, device: Union[str, torch.device] = "cuda"):
        if compute_dtype not in (BF16, torch.float16):
            raise NotImplementedError("VAE compute dtype is bfloat16 or float16 (fp32 accumulation)")
        self.compute_dtype = compute_dtype
        self._L = nv.lib(compute_dtype)
        self.timestep_conditioning = timestep_conditioning
        self.base_channels = base_channels
        self.decode_noise_scale = 0.025
        self.latent_channels = 128
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SimpleVideoDecoder runs on the MI355X only (no CPU fallback)")
        blocks = decoder_blocks if decoder_blocks is not None else _DEFAULT_DECODER_BLOCKS
        # up_blocks = reversed(decoder_blocks) (simple_decoder.py:403)
        self.plan: List[Tuple[str, dict, int]] = []
        ch = base_channels * 8
        for name, params in reversed(blocks):
            p = {"num_layers": params} if isinstance(params, int) else dict(params)
            if name == "res_x":
                self.plan.append(("res", p, ch))
            elif name in _STRIDE_MAP:
                q = {"stride": _STRIDE_MAP[name], "multiplier": p.get("multiplier", 1), "residual": p.get("residual", False)}
                self.plan.append(("upsample", q, ch))
                ch //= q["multiplier"]
            else:
                raise ValueError(f"Unknown decoder block: {name}")
        self.final_channels = ch
        self.timestep_scale_multiplier = 1000.0
        self._w: Dict[str, torch.Tensor] = {}
        self._h = None
        self._ws: Optional[torch.Tensor] = None
        self._ws_bytes = 0
        self.generator: Optional[torch.Generator] = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.ltx2_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def expected_weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}

        def conv(n, co, ci):
            s[n + ".weight"] = (co, ci, 3, 3, 3)
            s[n + ".bias"] = (co,)

        s["vae.per_channel_statistics.mean-of-means"] = (self.latent_channels,)
        s["vae.per_channel_statistics.std-of-means"] = (self.latent_channels,)
        conv("vae.decoder.conv_in.conv", self.base_channels * 8, self.latent_channels)
        for i, (kind, p, ch) in enumerate(self.plan):
            pre = f"vae.decoder.up_blocks.{i}"
            if kind == "res":
                for j in range(p["num_layers"]):
                    conv(f"{pre}.res_blocks.{j}.conv1.conv", ch, ch)
                    conv(f"{pre}.res_blocks.{j}.conv2.conv", ch, ch)
                    s[f"{pre}.res_blocks.{j}.scale_shift_table"] = (4, ch)
            else:
                conv(f"{pre}.conv.conv", math.prod(p["stride"]) * ch // p["multiplier"], ch)
        conv("vae.decoder.conv_out.conv", 48, self.final_channels)
        s["vae.decoder.last_scale_shift_table"] = (2, self.final_channels)
        return s

    def _register(self, name: str, t: torch.Tensor) -> None:
        t = t.contiguous()
        self._w[name] = t
        dt = nv.DTYPE_BF16 if t.dtype == self.compute_dtype else nv.DTYPE_F32
        nv.check(self._L.ltx2_vae_set_weight(self._h, name.encode(), nv.ptr(t), dt, t.numel()))

    def _create(self) -> None:
        if self._h is not None:
            self._L.ltx2_vae_destroy(self._h)
        cfg = nv.VaeConfig()
        cfg.n_blocks = len(self.plan)
        for i, (kind, p, ch) in enumerate(self.plan):
            cfg.kind[i] = nv.VAE_RES if kind == "res" else nv.VAE_UPSAMPLE
            cfg.num_layers[i] = p.get("num_layers", 0)
            st = p.get("stride", (1, 1, 1))
            for a in range(3):
                cfg.stride[i][a] = st[a]
            cfg.multiplier[i] = p.get("multiplier", 1)
            cfg.residual[i] = int(p.get("residual", False))
        cfg.base_channels = self.base_channels
        cfg.latent_channels = self.latent_channels
        cfg.timestep_conditioning = int(self.timestep_conditioning)
        cfg.decode_noise_scale = self.decode_noise_scale
        h = C.c_void_p()
        nv.check(self._L.ltx2_vae_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Checkpoint-keyed tensors (reference load_vae_decoder_weights key scheme,
        simple_decoder.py:592-671).  Conv weights (Cout,Cin,3,3,3) -> bf16 [Cout][27][Cin]
        (depth-to-space convs row-permuted); linears -> bf16; everything else fp32."""
        self._create()
        exp = self.expected_weight_shapes()
        for k, shp in exp.items():
            if k not in sd:
                raise KeyError(f"missing VAE weight {k}")
            if tuple(sd[k].shape) != shp:
                raise ValueError(f"VAE weight {k}: shape {tuple(sd[k].shape)} != expected {shp}")
        dev = self.device
        d2s = {f"vae.decoder.up_blocks.{i}.conv.conv": p["stride"] for i, (kind, p, _) in enumerate(self.plan) if kind == "upsample"}
        for k in exp:
            t = sd[k].to(dev, torch.float32)
            base = k.rsplit(".", 1)[0]
            if t.dim() == 5:
                self._register(k, K.conv_weight_to_engine(t, d2s.get(base), dtype=self.compute_dtype))
            elif k.endswith(".bias") and base in d2s:
                self._register(k, K.conv_bias_to_engine(t, d2s[base]))
            else:
                self._register(k, t.reshape(-1) if t.dim() <= 1 else t)
        # optional timestep-conditioning parameters (created at load time in the reference too)
        if self.timestep_conditioning:
            if "vae.decoder.timestep_scale_multiplier" in sd:
                self.timestep_scale_multiplier = float(sd["vae.decoder.timestep_scale_multiplier"])
            nv.check(self._L.ltx2_vae_set_timestep_multiplier(self._h, self.timestep_scale_multiplier))
            for k, t in sd.items():
                if ".time_embedder.timestep_embedder." in k or ".last_time_embedder.timestep_embedder." in k:
                    t = t.to(dev, torch.float32)
                    self._register(k, t.to(self.compute_dtype) if k.endswith(".weight") else t)

    @property
    def per_channel_statistics(self) -> "PerChannelStatistics":
        """Latent-space statistics shipped with the VAE weights (video_vae/ops.py:133-210)."""
        return PerChannelStatistics(self._w["vae.per_channel_statistics.mean-of-means"], self._w["vae.per_channel_statistics.std-of-means"])

    def init_random_weights(self, seed: int = 0) -> None:
        """Synthetic weights generated in HBM in engine layout (bench / smoke)."""
        self._create()
        g = torch.Generator(device=self.device).manual_seed(seed)
        dev = self.device

        def rn(*shape, scale=1.0):
            return torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * scale

        def conv(name, co, ci):
            self._register(name + ".weight", rn(co, 27, ci, scale=1.0 / math.sqrt(27 * ci)).to(self.compute_dtype))
            self._register(name + ".bias", rn(co, scale=0.02))

        def lin(name, o, i):
            self._register(name + ".weight", rn(o, i, scale=1.0 / math.sqrt(i)).to(self.compute_dtype))
            self._register(name + ".bias", rn(o, scale=0.02))

        self._register("vae.per_channel_statistics.mean-of-means", torch.zeros(self.latent_channels, device=dev))
        self._register("vae.per_channel_statistics.std-of-means", torch.ones(self.latent_channels, device=dev))
        conv("vae.decoder.conv_in.conv", self.base_channels * 8, self.latent_channels)
        for i, (kind, p, ch) in enumerate(self.plan):
            pre = f"vae.decoder.up_blocks.{i}"
            if kind == "res":
                for j in range(p["num_layers"]):
                    conv(f"{pre}.res_blocks.{j}.conv1.conv", ch, ch)
                    conv(f"{pre}.res_blocks.{j}.conv2.conv", ch, ch)
                    self._register(f"{pre}.res_blocks.{j}.scale_shift_table", rn(4, ch, scale=0.1))
                if self.timestep_conditioning:
                    lin(f"{pre}.time_embedder.timestep_embedder.linear_1", 4 * ch, 256)
                    lin(f"{pre}.time_embedder.timestep_embedder.linear_2", 4 * ch, 4 * ch)
            else:
                conv(f"{pre}.conv.conv", math.prod(p["stride"]) * ch // p["multiplier"], ch)
        conv("vae.decoder.conv_out.conv", 48, self.final_channels)
        self._register("vae.decoder.last_scale_shift_table", rn(2, self.final_channels, scale=0.1))
        if self.timestep_conditioning:
            lin("vae.decoder.last_time_embedder.timestep_embedder.linear_1", 256, 256)
            lin("vae.decoder.last_time_embedder.timestep_embedder.linear_2", 2 * self.final_channels, 256)
            nv.check(self._L.ltx2_vae_set_timestep_multiplier(self._h, self.timestep_scale_multiplier))

    # ------------------------------------------------------------------ decode
    def _bind(self, t: int, h: int, w: int) -> None:
        need = self._L.ltx2_vae_workspace_bytes(self._h, t, h, w)
        if need <= 0:
            raise ValueError(f"bad latent grid {t}x{h}x{w}")
        if self._ws is None or self._ws_bytes < need:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
            base = (self._ws.data_ptr() + 255) // 256 * 256
            nv.check(self._L.ltx2_vae_bind_workspace(self._h, base, need))

    def out_frames(self, t: int) -> int:
        return self._L.ltx2_vae_out_frames(self._h, t)

    def __call__(self, latent: torch.Tensor, timestep: Optional[float] = 0.05, show_progress: bool = True,
                 causal: bool = False, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """latent (B,128,T,H,W) -> video (B,3,T_px,H*32,W*32) fp32 in [-1,1]   (simple_decoder.py:446-563).

        `noise`: the N(0,1) tensor mixed in when timestep conditioning is active
        (x = noise*0.025 + 0.975*x, :496-498); drawn from torch's RNG when not supplied."""
        if self._h is None:
            raise RuntimeError("decoder weights not loaded")
        if latent.dim() != 5 or latent.shape[0] != 1:
            raise ValueError("latent must be (1, C, T, H, W)")
        _, c, t, h, w = latent.shape
        lat = latent[0].to(self.device, torch.float32).contiguous()
        tcond = self.timestep_conditioning and timestep is not None
        if tcond and noise is None:
            noise = torch.randn(lat.shape, generator=self.generator, device=self.device, dtype=torch.float32)
        nz = noise.reshape(lat.shape).to(self.device, torch.float32).contiguous() if (tcond and noise is not None) else None
        self._bind(t, h, w)
        tp = self.out_frames(t)
        sh = sw = 1
        for kind, p, _ in self.plan:
            if kind == "upsample":
                sh *= p["stride"][1]
                sw *= p["stride"][2]
        video = torch.empty(3, tp, h * sh * 4, w * sw * 4, device=self.device, dtype=torch.float32)
        nv.check(self._L.ltx2_vae_decode(self._h, nv.ptr(lat), t, h, w, float(timestep) if tcond else -1.0, nv.ptr(nz),
                                          int(causal), nv.ptr(video), nv.stream()))
        return video[None]


class PerChannelStatistics:
    """mean-of-means / std-of-means of the video VAE latent space (reference video_vae/ops.py:133-210)."""

    def __init__(self, mean_of_means: torch.Tensor, std_of_means: torch.Tensor):
        self.mean_of_means, self.std_of_means = mean_of_means.float(), std_of_means.float()

    def un_normalize(self, x: torch.Tensor) -> torch.Tensor:
        return x * self.std_of_means.reshape(1, -1, 1, 1, 1).to(x.device) + self.mean_of_means.reshape(1, -1, 1, 1, 1).to(x.device)

    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean_of_means.reshape(1, -1, 1, 1, 1).to(x.device)) / self.std_of_means.reshape(1, -1, 1, 1, 1).to(x.device)


def load_vae_decoder_weights(decoder: SimpleVideoDecoder, weights_path: str) -> None:
    """Load `vae.*` tensors from a safetensors checkpoint (reference simple_decoder.py:566-673)."""
    from safetensors import safe_open
    sd = {}
    with safe_open(weights_path, framework="pt") as f:
        for k in f.keys():
            if k.startswith("vae.decoder.") or k.startswith("vae.per_channel_statistics."):
                sd[k] = f.get_tensor(k)
    decoder.load_state_dict(sd)


def _latent_t_to_pixel_t(lt: int) -> int:
    for _ in range(3):
        lt = lt * 2 - 1
    return lt


def decode_latent(latent: torch.Tensor, decoder: SimpleVideoDecoder, timestep: Optional[float] = 0.05, key=None,
                  temporal_chunk_size: int = 7, temporal_overlap: int = 2,
                  noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Latent -> uint8 frames (T,H,W,3).  Same temporal chunking (7 latent frames, overlap 2),
    linear cross-fade and trim as reference simple_decoder.py:676-800, so results agree frame for
    frame with the reference's chunked decode."""
    if latent.dim() == 4:
        latent = latent[None]
    T = latent.shape[2]
    sl = (lambda s, e: None) if noise is None else (lambda s, e: noise.reshape(latent.shape)[:, :, s:e])
    if T <= temporal_chunk_size:
        video = decoder(latent, timestep=timestep, noise=noise)
    else:
        total = _latent_t_to_pixel_t(T)
        stride = temporal_chunk_size - temporal_overlap
        chunks = []
        t = 0
        while t < T:
            end = min(t + temporal_chunk_size, T)
            if end - t < temporal_overlap + 1 and t > 0:
                t = max(0, end - temporal_chunk_size)
                end = min(t + temporal_chunk_size, T)
            chunks.append(decoder(latent[:, :, t:end], timestep=timestep, noise=sl(t, end)))
            if end >= T:
                break
            t += stride
        if len(chunks) == 1:
            video = chunks[0][:, :, :total]
        else:
            # cross-fade, trim and uint8 conversion fused per chunk (the reference concatenates / blends fp32 volumes, :760-798):
            # chunk i starts `ov` frames before the end of what precedes it; its first ov frames are blended with that tail
            ov_ref = _latent_t_to_pixel_t(temporal_overlap)
            _, _, _, Hp, Wp = chunks[0].shape
            frames = torch.empty(total, Hp, Wp, 3, device=chunks[0].device, dtype=torch.uint8)
            K.video_chunk_to_uint8(chunks[0][0], frames, 0)
            length = chunks[0].shape[2]
            prev = chunks[0]
            for cur in chunks[1:]:
                ov = min(ov_ref, cur.shape[2], length)
                if ov <= 1:
                    K.video_chunk_to_uint8(cur[0], frames, length)
                    length += cur.shape[2]
                else:
                    if ov > prev.shape[2]:
                        raise ValueError("temporal overlap longer than the previous chunk")
                    ramp = torch.linspace(0.0, 1.0, ov, device=cur.device)
                    K.video_chunk_to_uint8(cur[0], frames, length - ov, prev=prev[0], ramp=ramp)
                    length += cur.shape[2] - ov
                prev = cur
            return frames
    return K.video_to_uint8(video[0])


# ---------------------------------------------------------------------------------------------
# Tiled decode (reference video_vae/tiling.py)
# ---------------------------------------------------------------------------------------------
def compute_trapezoidal_mask_1d(length: int, ramp_left: int, ramp_right: int, left_starts_from_0: bool = False,
                                device=None) -> torch.Tensor:
    if length <= 0:
        raise ValueError("Mask length must be positive.")
    ramp_left = max(0, min(ramp_left, length))
    ramp_right = max(0, min(ramp_right, length))
    mask = torch.ones(length, device=device)
    if ramp_left > 0:
        n = ramp_left + 1 if left_starts_from_0 else ramp_left + 2
        fade = torch.linspace(0.0, 1.0, n, device=device)[:-1]
        if not left_starts_from_0:
            fade = fade[1:]
        mask = torch.cat([fade, mask[ramp_left:]])
    if ramp_right > 0:
        mask = torch.cat([mask[:-ramp_right], torch.linspace(1.0, 0.0, ramp_right + 2, device=device)[1:-1]])
    return mask.clamp(0, 1)


@dataclass(frozen=True)
class SpatialTilingConfig:
    tile_size_in_pixels: int
    tile_overlap_in_pixels: int = 0

    def __post_init__(self) -> None:
        if self.tile_size_in_pixels < 64:
            raise ValueError(f"tile_size_in_pixels must be at least 64, got {self.tile_size_in_pixels}")
        if self.tile_size_in_pixels % 32 != 0:
            raise ValueError(f"tile_size_in_pixels must be divisible by 32, got {self.tile_size_in_pixels}")
        if self.tile_overlap_in_pixels % 32 != 0:
            raise ValueError(f"tile_overlap_in_pixels must be divisible by 32, got {self.tile_overlap_in_pixels}")
        if self.tile_overlap_in_pixels >= self.tile_size_in_pixels:
            raise ValueError(f"Overlap must be less than tile size, got {self.tile_overlap_in_pixels} and {self.tile_size_in_pixels}")


@dataclass(frozen=True)
class TemporalTilingConfig:
    tile_size_in_frames: int
    tile_overlap_in_frames: int = 0

    def __post_init__(self) -> None:
        if self.tile_size_in_frames < 16:
            raise ValueError(f"tile_size_in_frames must be at least 16, got {self.tile_size_in_frames}")
        if self.tile_size_in_frames % 8 != 0:
            raise ValueError(f"tile_size_in_frames must be divisible by 8, got {self.tile_size_in_frames}")
        if self.tile_overlap_in_frames % 8 != 0:
            raise ValueError(f"tile_overlap_in_frames must be divisible by 8, got {self.tile_overlap_in_frames}")
        if self.tile_overlap_in_frames >= self.tile_size_in_frames:
            raise ValueError(f"Overlap must be less than tile size, got {self.tile_overlap_in_frames} and {self.tile_size_in_frames}")


@dataclass(frozen=True)
class TilingConfig:
    spatial_config: Optional[SpatialTilingConfig] = None
    temporal_config: Optional[TemporalTilingConfig] = None

    @classmethod
    def default(cls) -> "TilingConfig":
        return cls(spatial_config=SpatialTilingConfig(512, 64), temporal_config=TemporalTilingConfig(64, 24))


@dataclass
class TileSpec:
    in_t_start: int
    in_t_end: int
    in_h_start: int
    in_h_end: int
    in_w_start: int
    in_w_end: int
    out_t_start: int
    out_t_end: int
    out_h_start: int
    out_h_end: int
    out_w_start: int
    out_w_end: int
    ramp_t_left: int
    ramp_t_right: int
    ramp_h_left: int
    ramp_h_right: int
    ramp_w_left: int
    ramp_w_right: int


def _tiles_1d(length: int, tile: int, overlap: int) -> List[Tuple[int, int, int, int]]:
    if length <= tile:
        return [(0, length, 0, 0)]
    out, pos, stride = [], 0, tile - overlap
    while pos < length:
        end = min(pos + tile, length)
        start = max(0, end - tile)
        out.append((start, end, overlap if start > 0 else 0, overlap if end < length else 0))
        if end >= length:
            break
        pos += stride
    return out


def generate_tile_specs(latent_shape, tiling_config: TilingConfig, scale_factors=(8, 32, 32)) -> List[TileSpec]:
    _, _, t, h, w = latent_shape
    st, sh, sw = scale_factors
    sc, tc = tiling_config.spatial_config, tiling_config.temporal_config
    th, oh = (sc.tile_size_in_pixels // sh, sc.tile_overlap_in_pixels // sh) if sc else (h, 0)
    tw, ow = (sc.tile_size_in_pixels // sw, sc.tile_overlap_in_pixels // sw) if sc else (w, 0)
    tt, ot = (tc.tile_size_in_frames // st, tc.tile_overlap_in_frames // st) if tc else (t, 0)
    specs = []
    for t0, t1, tl, tr in _tiles_1d(t, tt, ot):
        for h0, h1, hl, hr in _tiles_1d(h, th, oh):
            for w0, w1, wl, wr in _tiles_1d(w, tw, ow):
                specs.append(TileSpec(t0, t1, h0, h1, w0, w1,
                                      t0 * st if t0 > 0 else 0, (t1 - 1) * st + 1 if t1 > 1 else 1,
                                      h0 * sh, h1 * sh, w0 * sw, w1 * sw,
                                      tl * st, tr * st, hl * sh, hr * sh, wl * sw, wr * sw))
    return specs


def decode_tiled(latent: torch.Tensor, decoder_fn: Callable, tiling_config: TilingConfig, timestep: Optional[float] = 0.05,
                 show_progress: bool = True, key=None) -> Iterator[torch.Tensor]:
    """Trapezoid-blended tiled decode; yields the blended (B,3,T,H,W) video once
    (reference tiling.py:252-412; its dead first loop, which decodes every tile twice, is not
    reproduced -- it has no effect on the result)."""
    b, c, t, h, w = latent.shape
    out_t, out_h, out_w = (t - 1) * 8 + 1, h * 32, w * 32
    dev = latent.device
    if b != 1:
        raise ValueError("batch must be 1")
    output = torch.zeros(3, out_t, out_h, out_w, device=dev)
    weights = torch.zeros(out_t, out_h, out_w, device=dev)
    for s in generate_tile_specs(latent.shape, tiling_config):
        tile = decoder_fn(latent[:, :, s.in_t_start:s.in_t_end, s.in_h_start:s.in_h_end, s.in_w_start:s.in_w_end], timestep=timestep)
        _, _, dt, dh, dw = tile.shape
        nt, nh, nw = min(dt, s.out_t_end - s.out_t_start), min(dh, s.out_h_end - s.out_h_start), min(dw, s.out_w_end - s.out_w_start)
        mt = compute_trapezoidal_mask_1d(nt, min(s.ramp_t_left, nt), min(s.ramp_t_right, nt), left_starts_from_0=(s.out_t_start == 0), device=dev)
        mh = compute_trapezoidal_mask_1d(nh, min(s.ramp_h_left, nh), min(s.ramp_h_right, nh), device=dev)
        mw = compute_trapezoidal_mask_1d(nw, min(s.ramp_w_left, nw), min(s.ramp_w_right, nw), device=dev)
        # output += tile * mask, weights += mask in one pass over the tile (blend-accumulate kernel)
        K.tile_blend_accumulate(tile[0].to(dev), nt, nh, nw, mt, mh, mw, output, weights, s.out_t_start, s.out_h_start, s.out_w_start)
    K.tile_blend_finish(output, weights)
    yield output[None]
