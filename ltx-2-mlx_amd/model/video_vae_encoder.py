"""Video VAE encoder on MI355X behind the reference's API (LTX_2_MLX/model/video_vae/simple_encoder.py:258-411
SimpleVideoEncoder, :414-520 load_vae_encoder_weights): image / video in [-1, 1] -> normalised latent, used by
image conditioning (pipelines/common.py:105-146).

Convs run on the implicit-GEMM MFMA kernels (`ltx2_conv3d_fused` with pad_zero=2: zero padding in H/W, causal
replicated temporal edge); pixel_norm + SiLU is `ltx2_pixnorm_mod_silu` with a zero table; the
SpaceToDepthDownsample3d tail (space-to-depth + group-mean residual) is `ltx2_s2d_downsample`.  Activations are
channels-last bf16.  Two channel counts are padded for the 16-byte / power-of-two kernels: the 48 patchified
input channels to 64 (zero weights), and the 129 conv_out rows to 132 (the extra rows are dropped)."""
from __future__ import annotations

from typing import Dict, List, Tuple, Union

import torch

from .. import kernels as K
from .video_vae import PerChannelStatistics

BF16 = torch.bfloat16
# (kind, arg): res = number of blocks; down = (out_channels, stride)   (reference simple_encoder.py:293-306)
ENCODER_BLOCKS = [("res", 4), ("down", (256, (1, 2, 2))), ("res", 6), ("down", (512, (2, 1, 1))), ("res", 6),
                  ("down", (1024, (2, 2, 2))), ("res", 2), ("down", (1024, (2, 2, 2))), ("res", 2)]


def patchify_video(video: torch.Tensor, q: int = 4, pad_to: int = 64) -> torch.Tensor:
    """(3, F, H, W) fp32 -> channels-last bf16 [F, H/q, W/q, pad_to]; channel = (c*q + r_w)*q + r_h
    (reference video_vae/ops.py:44-60), zero-padded to `pad_to` channels.  Layout glue on torch tensors."""
    c, f, h, w = video.shape
    x = video.reshape(c, f, h // q, q, w // q, q).permute(1, 2, 4, 0, 5, 3).reshape(f, h // q, w // q, c * q * q)
    out = torch.zeros(f, h // q, w // q, pad_to, device=video.device, dtype=BF16)
    out[..., :c * q * q] = x.to(BF16)
    return out


class SimpleVideoEncoder:
    def __init__(self, compute_dtype: torch.dtype = BF16, device: Union[str, torch.device] = "cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SimpleVideoEncoder runs on the MI355X only (no CPU fallback); got device " + str(device))
        self.patch_size = 4
        self.blocks = ENCODER_BLOCKS
        self._w: Dict[str, torch.Tensor] = {}
        self.per_channel_statistics = PerChannelStatistics(torch.zeros(128), torch.ones(128))
        self._zero_tab: Dict[int, torch.Tensor] = {}
        self._loaded = False

    # ------------------------------------------------------------------ weights
    def expected_weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}

        def conv(name, co, ci):
            s[name + ".weight"] = (co, ci, 3, 3, 3)
            s[name + ".bias"] = (co,)

        conv("vae.encoder.conv_in.conv", 128, 48)
        ch = 128
        for i, (kind, arg) in enumerate(self.blocks):
            p = f"vae.encoder.down_blocks.{i}"
            if kind == "res":
                for j in range(arg):
                    conv(f"{p}.res_blocks.{j}.conv1.conv", ch, ch)
                    conv(f"{p}.res_blocks.{j}.conv2.conv", ch, ch)
            else:
                out, st = arg
                conv(f"{p}.conv.conv", out // (st[0] * st[1] * st[2]), ch)
                ch = out
        conv("vae.encoder.conv_out.conv", 129, ch)
        s["vae.per_channel_statistics.mean-of-means"] = (128,)
        s["vae.per_channel_statistics.std-of-means"] = (128,)
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        exp = self.expected_weight_shapes()
        missing = [k for k in exp if k not in sd]
        if missing and strict:
            raise KeyError(f"missing {len(missing)} encoder weights, e.g. {missing[:4]}")
        dev = self.device
        for k, shp in exp.items():
            if k not in sd:
                continue
            t = sd[k].to(dev)
            if tuple(t.shape) != shp:
                raise ValueError(f"weight {k}: shape {tuple(t.shape)} != expected {shp}")
            if k == "vae.encoder.conv_in.conv.weight":            # 48 -> 64 input channels (zeros)
                t = torch.cat([t.float(), torch.zeros(t.shape[0], 16, 3, 3, 3, device=dev)], dim=1)
            if k.startswith("vae.encoder.conv_out.conv."):        # 129 -> 132 output rows (dropped after the conv)
                t = torch.cat([t.float(), torch.zeros((3,) + tuple(t.shape[1:]), device=dev)], dim=0)
            self._w[k] = K.conv_weight_to_engine(t) if t.dim() == 5 else t.float().contiguous()
        self.per_channel_statistics = PerChannelStatistics(self._w["vae.per_channel_statistics.mean-of-means"],
                                                           self._w["vae.per_channel_statistics.std-of-means"])
        self._loaded = True

    def init_random_weights(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shp in self.expected_weight_shapes().items():
            if k.endswith("mean-of-means"):
                sd[k] = torch.zeros(shp)
            elif k.endswith("std-of-means"):
                sd[k] = torch.ones(shp)
            elif k.endswith(".bias"):
                sd[k] = 0.02 * torch.randn(shp, generator=g)
            else:
                sd[k] = torch.randn(shp, generator=g) / (27 * shp[1]) ** 0.5
        self.load_state_dict(sd)

    # ------------------------------------------------------------------ forward
    def _norm_act(self, x: torch.Tensor) -> torch.Tensor:
        """pixel_norm over channels + SiLU, no affine (reference simple_encoder.py:10-13,143-150)."""
        C = x.shape[-1]
        if C not in self._zero_tab:
            self._zero_tab[C] = torch.zeros(2, C, device=self.device, dtype=torch.float32)
        return K.pixnorm_mod_silu(x, self._zero_tab[C], None, shift_row=0, scale_row=1)

    def _conv(self, x: torch.Tensor, name: str) -> torch.Tensor:
        return K.conv3d(x, self._w[name + ".weight"], self._w[name + ".bias"], causal=True, pad_zero=2)

    def __call__(self, video: torch.Tensor, show_progress: bool = True) -> torch.Tensor:
        """video (B, 3, F, H, W) in [-1, 1], F = 1 + 8k -> normalised latent (B, 128, 1 + k, H/32, W/32) fp32."""
        if not self._loaded:
            raise RuntimeError("SimpleVideoEncoder: weights not loaded")
        if video.dim() != 5 or video.shape[0] != 1 or video.shape[1] != 3:
            raise ValueError(f"expected (1, 3, F, H, W), got {tuple(video.shape)}")
        f, h, w = video.shape[2:]
        if (f - 1) % 8 != 0:
            raise ValueError(f"Invalid number of frames: {f}. Encoder input must have 1 + 8*k frames (e.g., 1, 9, 17, 25, 33...).")
        if h % 32 or w % 32:
            raise ValueError(f"Resolution ({h}x{w}) must be divisible by 32")
        x = patchify_video(video[0].to(self.device, torch.float32))
        x = self._conv(x, "vae.encoder.conv_in.conv")
        for i, (kind, arg) in enumerate(self.blocks):
            p = f"vae.encoder.down_blocks.{i}"
            if kind == "res":
                for j in range(arg):
                    hcur = self._conv(self._norm_act(x), f"{p}.res_blocks.{j}.conv1.conv")
                    x = K.conv3d(self._norm_act(hcur), self._w[f"{p}.res_blocks.{j}.conv2.conv.weight"],
                                 self._w[f"{p}.res_blocks.{j}.conv2.conv.bias"], causal=True, mode=1, res=x, pad_zero=2)
            else:
                _, st = arg
                if st[0] == 2:                      # duplicate the first frame (reference :230-232)
                    x = torch.cat([x[:1], x], dim=0)
                x = K.s2d_downsample(self._conv(x, p + ".conv.conv"), x, st)
        x = self._conv(self._norm_act(x), "vae.encoder.conv_out.conv")          # [T', H', W', 132]
        means = x[..., :128].contiguous()
        st = self.per_channel_statistics
        return K.latent_normalize_nchw(means, st.mean_of_means, st.std_of_means)[None]


def load_vae_encoder_weights(encoder: SimpleVideoEncoder, weights_path: str) -> None:
    """Load `vae.encoder.*` / `vae.per_channel_statistics.*` from safetensors (reference simple_encoder.py:414-520)."""
    from safetensors import safe_open
    sd = {}
    with safe_open(weights_path, framework="pt") as f:
        for k in f.keys():
            if k.startswith("vae.encoder.") or k.startswith("vae.per_channel_statistics."):
                sd[k] = f.get_tensor(k)
    encoder.load_state_dict(sd)


def encode_video(video: torch.Tensor, encoder: SimpleVideoEncoder) -> torch.Tensor:
    """(T, H, W, 3) uint8 or float in [0, 1] / (B, 3, T, H, W) in [-1, 1] -> latent (reference simple_encoder.py:523-564)."""
    if video.dim() == 4:
        v = video.float()
        if video.dtype == torch.uint8:
            v = v / 255.0
        video = (v * 2 - 1).permute(3, 0, 1, 2)[None]
    return encoder(video)
