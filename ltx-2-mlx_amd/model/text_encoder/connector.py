"""Embeddings1DConnector on MI355X behind the reference's API
(LTX_2_MLX/model/text_encoder/connector.py:13-101 BasicTransformerBlock1D, :104-283 Embeddings1DConnector).

Per prompt: the (B, T, 3840) text features are extended with the tiled learnable registers to >= 1024 tokens, run
through `num_layers` pre-norm blocks (weightless RMSNorm -> self-attention with q/k RMSNorm and INTERLEAVED RoPE over
the token index -> residual; RMSNorm -> GELU-tanh FFN -> residual) and a final RMSNorm.  Everything heavy is the
library's kernels: `ltx2_adaln_rmsnorm`, fused QKV `ltx2_gemm_bf16`, `ltx2_qknorm_rope`, `ltx2_vt_transpose`,
`ltx2_flash_attn` (30 heads x 128, no mask once the registers are in), the residual-accumulating GEMM epilogue.

INTERLEAVED RoPE without a second rotation kernel: attention only sees q.k per head, so any permutation of a head's
dimensions applied to BOTH q and k leaves it unchanged.  At load time the rows of to_q / to_k (and their biases and
q_norm / k_norm weights) are permuted per head to [even dims | odd dims]; the reference's pair (2j, 2j+1) then sits at
(j, j + 64) of the head, which is exactly the SPLIT layout `ltx2_qknorm_rope` rotates, with the table slot h*64 + j
holding the frequency of interleaved pair h*64 + j (`ltx2_rope_tables` on the 1-D index grid)."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from ... import _native as nv
from ... import kernels as K

BF16 = torch.bfloat16


def _split_perm(heads: int, head_dim: int, device) -> torch.Tensor:
    """new row h*hd + j <- old row h*hd + 2j (j < hd/2), new row h*hd + hd/2 + j <- old row h*hd + 2j + 1."""
    j = torch.arange(head_dim // 2, device=device)
    one = torch.cat([2 * j, 2 * j + 1])
    return (torch.arange(heads, device=device)[:, None] * head_dim + one[None, :]).reshape(-1)


class Embeddings1DConnector:
    """Constructor keywords as the reference (connector.py:114-126).  `rope_type` other than interleaved, and a run
    without learnable registers on a padded prompt (which would need an attention mask), are rejected."""

    def __init__(self, attention_head_dim: int = 128, num_attention_heads: int = 30, num_layers: int = 2,
                 positional_embedding_theta: float = 10000.0, positional_embedding_max_pos: Optional[List[int]] = None,
                 num_learnable_registers: Optional[int] = 128, rope_type: str = "interleaved", norm_eps: float = 1e-6,
                 apply_gated_attention: bool = False, double_precision_rope: bool = False,
                 device: Union[str, torch.device] = "cuda"):
        if str(getattr(rope_type, "value", rope_type)).lower() != "interleaved":
            raise NotImplementedError("Embeddings1DConnector: only the INTERLEAVED RoPE of the released checkpoints is implemented")
        if attention_head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 128 or 64 (flash-attention kernel instantiations)")
        self.num_attention_heads, self.attention_head_dim, self.num_layers = num_attention_heads, attention_head_dim, num_layers
        self.inner_dim = num_attention_heads * attention_head_dim
        self.positional_embedding_theta = positional_embedding_theta
        self.positional_embedding_max_pos = positional_embedding_max_pos or [1]
        self.num_learnable_registers = num_learnable_registers
        self.norm_eps, self.apply_gated_attention, self.double_precision_rope = norm_eps, apply_gated_attention, double_precision_rope
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Embeddings1DConnector runs on the MI355X only (no CPU fallback); got device " + str(device))
        self.learnable_registers: Optional[torch.Tensor] = None
        self._blocks: List[Dict[str, torch.Tensor]] = []
        self._tables: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._loaded = False

    # ------------------------------------------------------------------ weights
    def expected_weight_shapes(self) -> Dict[str, tuple]:
        d = self.inner_dim
        s: Dict[str, tuple] = {}
        if self.num_learnable_registers:
            s["learnable_registers"] = (self.num_learnable_registers, d)
        for i in range(self.num_layers):
            p = f"transformer_1d_blocks.{i}"
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                s[f"{p}.attn1.{n}.weight"], s[f"{p}.attn1.{n}.bias"] = (d, d), (d,)
            s[f"{p}.attn1.q_norm.weight"] = s[f"{p}.attn1.k_norm.weight"] = (d,)
            if self.apply_gated_attention:
                s[f"{p}.attn1.to_gate_logits.weight"], s[f"{p}.attn1.to_gate_logits.bias"] = (self.num_attention_heads, d), (self.num_attention_heads,)
            s[f"{p}.ff.net.0.proj.weight"], s[f"{p}.ff.net.0.proj.bias"] = (4 * d, d), (4 * d,)
            s[f"{p}.ff.net.2.weight"], s[f"{p}.ff.net.2.bias"] = (d, 4 * d), (d,)
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """Checkpoint keys below `model.diffusion_model.video_embeddings_connector.` (reference encoder.py:452-520).
        Linear weights stay [out, in] and become bf16; to_q / to_k / to_v are fused into one projection with the
        q and k rows permuted per head (module docstring); biases and norm weights become fp32."""
        exp = self.expected_weight_shapes()
        missing = [k for k in exp if k not in sd]
        if missing and strict:
            raise KeyError(f"missing {len(missing)} connector weights, e.g. {missing[:4]}")
        for k, shp in exp.items():
            if k in sd and tuple(sd[k].shape) != shp:
                raise ValueError(f"weight {k}: shape {tuple(sd[k].shape)} != expected {shp}")
        dev = self.device
        perm = _split_perm(self.num_attention_heads, self.attention_head_dim, dev)

        def f(k):
            return sd[k].to(dev, torch.float32)

        if self.num_learnable_registers:
            self.learnable_registers = f("learnable_registers").contiguous()
        self._blocks = []
        for i in range(self.num_layers):
            p = f"transformer_1d_blocks.{i}"
            a = p + ".attn1."
            blk = {
                "wqkv": torch.cat([f(a + "to_q.weight")[perm], f(a + "to_k.weight")[perm], f(a + "to_v.weight")]).to(BF16).contiguous(),
                "bqkv": torch.cat([f(a + "to_q.bias")[perm], f(a + "to_k.bias")[perm], f(a + "to_v.bias")]).contiguous(),
                "qn": f(a + "q_norm.weight")[perm].contiguous(), "kn": f(a + "k_norm.weight")[perm].contiguous(),
                "wo": f(a + "to_out.0.weight").to(BF16).contiguous(), "bo": f(a + "to_out.0.bias").contiguous(),
                "w1": f(p + ".ff.net.0.proj.weight").to(BF16).contiguous(), "b1": f(p + ".ff.net.0.proj.bias").contiguous(),
                "w2": f(p + ".ff.net.2.weight").to(BF16).contiguous(), "b2": f(p + ".ff.net.2.bias").contiguous(),
            }
            if self.apply_gated_attention:
                blk["wg"] = f(a + "to_gate_logits.weight").to(BF16).contiguous()
                blk["bg"] = f(a + "to_gate_logits.bias").contiguous()
            self._blocks.append(blk)
        self._loaded = True

    def init_random_weights(self, seed: int = 0, std: float = 0.02) -> None:
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shp in self.expected_weight_shapes().items():
            if k == "learnable_registers":
                sd[k] = torch.rand(shp, generator=g) * 2 - 1
            elif k.endswith("_norm.weight"):
                sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
            else:
                sd[k] = std * torch.randn(shp, generator=g)
        self.load_state_dict(sd)

    # ------------------------------------------------------------------ forward
    def _rope_tables(self, seq_len: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos, sin fp32 [T, inner_dim/2]: slot p = interleaved pair p, angle grid[p] * (2 t / max_pos - 1)
        (reference rope.py:242-289,330-362 through connector.py:253-270)."""
        if seq_len not in self._tables:
            d = self.inner_dim
            if d % 2:
                raise ValueError("inner_dim must be even")
            n = d // 2
            if self.double_precision_rope:          # rope.py:147-178 (float64 power, cast to fp32)
                grid = torch.from_numpy((np.power(float(self.positional_embedding_theta), np.linspace(0.0, 1.0, n, dtype=np.float64))
                                         * math.pi / 2).astype(np.float32))
            else:                                   # rope.py:181-211
                grid = (torch.tensor(float(self.positional_embedding_theta)) ** torch.linspace(0.0, 1.0, n, dtype=torch.float32)
                        * (math.pi / 2)).float()
            idx = torch.arange(seq_len, dtype=torch.float32, device=self.device)
            pos = torch.stack([idx, idx], dim=-1)[None].contiguous()          # [1 dim, T, (start, end)] -> mid = idx
            mp = torch.tensor([float(self.positional_embedding_max_pos[0])], device=self.device)
            cos = torch.empty(seq_len, n, device=self.device, dtype=torch.float32)
            sin = torch.empty_like(cos)
            nv.check(nv.lib().ltx2_rope_tables(nv.ptr(pos), nv.ptr(grid.to(self.device)), nv.ptr(mp), seq_len, 1, n, n, nv.ptr(cos), nv.ptr(sin),
                                               nv.stream()))
            self._tables[seq_len] = (cos, sin)
        return self._tables[seq_len]

    def _append_learnable_registers(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """connector.py:173-230: extend to max(1024, T) rounded up to a multiple of the register count with the tiled
        registers' rows [T:]; original rows stay in place; the attention mask is cleared."""
        b, t, _ = hidden_states.shape
        n = self.num_learnable_registers
        dup = math.ceil(max(1024, t) / n)
        extra = self.learnable_registers.repeat(dup, 1)[t:]
        if extra.shape[0] == 0:
            return hidden_states
        return torch.cat([hidden_states, extra[None].expand(b, -1, -1)], dim=1)

    def __call__(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """hidden_states [B, T, inner_dim]; attention_mask additive [B,1,1,T] or None.  Returns (fp32 [B, T', inner_dim],
        additive mask of zeros [B,1,1,T']) like the reference (connector.py:232-283)."""
        if not self._loaded:
            raise RuntimeError("Embeddings1DConnector: weights not loaded")
        if hidden_states.dim() != 3 or hidden_states.shape[-1] != self.inner_dim:
            raise ValueError(f"hidden_states must be [B, T, {self.inner_dim}], got {tuple(hidden_states.shape)}")
        x = hidden_states.to(self.device, torch.float32)
        if self.num_learnable_registers:
            x = self._append_learnable_registers(x)
        elif attention_mask is not None and bool((attention_mask < -0.5).any()):
            raise NotImplementedError("a padded prompt without learnable registers needs a masked attention; not on this path")
        b, t, d = x.shape
        cos, sin = self._rope_tables(t)
        h, hd = self.num_attention_heads, self.attention_head_dim
        outs = []
        for bi in range(b):
            xs = x[bi].contiguous()                                  # fp32 residual stream [T, D]
            for blk in self._blocks:
                n1 = K.adaln_rmsnorm(xs, self.norm_eps)
                qkv = K.gemm(n1, blk["wqkv"], blk["bqkv"])            # [T, 3D] bf16
                K.qknorm_rope_(qkv, d, hd, 0, blk["qn"], d, blk["kn"], self.norm_eps, cos, sin)
                vt = K.vt_transpose(qkv[:, 2 * d:], h, hd)
                att = K.flash_attn(qkv[:, :d], qkv[:, d:2 * d], vt, h, t)
                if self.apply_gated_attention:
                    K.attn_head_gate_(att, n1, blk["wg"], blk["bg"], h)
                K.gemm(att, blk["wo"], blk["bo"], epilogue=nv.EPI_RESID_GATE_F32, out=xs)
                n2 = K.adaln_rmsnorm(xs, self.norm_eps)
                ff = K.gemm(n2, blk["w1"], blk["b1"], epilogue=nv.EPI_GELU_BF16)
                K.gemm(ff, blk["w2"], blk["b2"], epilogue=nv.EPI_RESID_GATE_F32, out=xs)
            outs.append(K.adaln_rmsnorm(xs, self.norm_eps).float())
        out = torch.stack(outs)
        return out, torch.zeros(b, 1, 1, t, device=self.device, dtype=torch.float32)
