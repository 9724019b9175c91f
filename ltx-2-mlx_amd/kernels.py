"""Tensor-level wrappers over the per-kernel C-ABI entry points (counterpart of the reference's
LTX_2_MLX/kernels/ + the mx.fast.* / mx.conv2d leaf calls).  Used by the unit-parity tests and
by host glue; the full DiT step and VAE pass go through the engine calls instead
(ltx2_dit_* / ltx2_vae_*), which sequence the same kernels in C++.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _native as nv

BF16 = torch.bfloat16
F16 = torch.float16
ACT16 = (BF16, F16)         # the 16-bit activation / weight type: bfloat16 (libltx2hip.so) or IEEE half (libltx2hip_f16.so, -DLTX2_F16)


def _L(*tensors, dtype=None):
    """The library build for these tensors' 16-bit type (the first bf16 / f16 tensor decides; `dtype` when there is none)."""
    for t in tensors:
        if t is not None and t.dtype in ACT16:
            return nv.lib(t.dtype)
    return nv.lib(dtype)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = nv.EPI_BF16,
         out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
         gate_table: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T + bias).  a, w bf16; bias/gate fp32."""
    assert a.dtype in ACT16 and w.dtype == a.dtype and a.dim() == 2 and w.dim() == 2
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        odt = torch.float32 if epilogue in (nv.EPI_F32, nv.EPI_RESID_GATE_F32) else a.dtype
        assert epilogue != nv.EPI_RESID_GATE_F32, "RESID_GATE accumulates into `out`; pass it"
        out = torch.empty(M, N, device=a.device, dtype=odt)
    gs = 0
    if gate is not None:
        gate = _c(gate)
        gs = 0 if gate.shape[0] == 1 else gate.stride(0)
    nv.check(_L(a).ltx2_gemm_bf16(nv.ptr(a), a.stride(0), nv.ptr(w), nv.ptr(bias), nv.ptr(out), out.stride(0), M, N, K,
                                     epilogue, nv.ptr(gate), gs, nv.ptr(gate_table), nv.ptr(res),
                                     res.stride(0) if res is not None else 0, nv.stream()))
    return out


def gemm_qkv_vt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], heads: int, head_dim: int = 128):
    """Fused QKV projection: returns (qkv [M, 3D] bf16 -- V columns only written on the unfused path --, vt [H, hd, Npad], fused)."""
    assert a.dtype in ACT16 and w.dtype == a.dtype
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    D = heads * head_dim
    assert N == 3 * D
    npad = (M + 63) // 64 * 64
    out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    vt = torch.empty(heads, head_dim, npad, device=a.device, dtype=a.dtype)
    import ctypes
    fused = ctypes.c_int(0)
    nv.check(_L(a).ltx2_gemm_qkv_vt(nv.ptr(a), a.stride(0), nv.ptr(w), nv.ptr(bias), nv.ptr(out), out.stride(0), M, N, K,
                                       nv.ptr(vt), 2 * D, npad, head_dim, ctypes.byref(fused), nv.stream()))
    return out, vt, bool(fused.value)


def adaln_rmsnorm2(x: torch.Tensor, scale0, shift0, scale1, shift1, eps: float = 1e-6, dtype: torch.dtype = BF16):
    """(rms_norm(x) * (1 + scale0) + shift0, rms_norm(x) * (1 + scale1) + shift1) from one read of x [rows, D] fp32."""
    x = _c(x.float())
    o0 = torch.empty(x.shape, device=x.device, dtype=dtype)
    o1 = torch.empty_like(o0)
    nv.check(nv.lib(dtype).ltx2_adaln_rmsnorm2(nv.ptr(x), x.stride(0), nv.ptr(o0), nv.ptr(o1), o0.stride(0), x.shape[0], x.shape[1], eps,
                                               nv.ptr(scale0), nv.ptr(shift0), nv.ptr(scale1), nv.ptr(shift1), nv.stream()))
    return o0, o1


def gemm_rowss(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """out = a @ w^T + bias (16-bit) plus the row partial sums of squares of out over 64-column strips -> (out, rowss [M, N/64] fp32 or None
    when the shape does not run on the kernel that writes them)."""
    assert a.dtype in ACT16 and w.dtype == a.dtype
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    rowss = torch.zeros(M, max(N // 64, 1), device=a.device, dtype=torch.float32)
    import ctypes
    written = ctypes.c_int(0)
    nv.check(_L(a).ltx2_gemm_bf16_rowss(nv.ptr(a), a.stride(0), nv.ptr(w), nv.ptr(bias), nv.ptr(out), out.stride(0), M, N, K, nv.ptr(rowss),
                                        ctypes.byref(written), nv.stream()))
    return out, (rowss if written.value else None)


def gemm_fold(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int, out: Optional[torch.Tensor] = None,
              gate_table: Optional[torch.Tensor] = None, shadow: Optional[torch.Tensor] = None, shadow_scale: Optional[torch.Tensor] = None,
              rf_parts: Optional[torch.Tensor] = None, rf_dim: int = 0, eps: float = 1e-6):
    """ltx2_gemm_bf16_fold (include/ltx2hip.h): the producer / consumer half of an RMS norm folded around two GEMMs.
    Producer (epilogue RESID_GATE_F32, `out` = the fp32 residual stream, updated in place; shadow = 16-bit [M, N]): returns (out, shadow_ss [N/256, ld]),
    the partial sums of squares of the new rows per 256-column tile, tile-major.
    Consumer (BF16 / GELU_BF16; rf_parts = a producer's shadow_ss over rf_dim columns): returns out [M, N].
    None when the 4-wave kernel does not take the problem."""
    import ctypes
    assert a.dtype in ACT16 and w.dtype == a.dtype
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    prod = epilogue == nv.EPI_RESID_GATE_F32
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    ss = torch.zeros(max(N // 256, 1), (M + 255) // 256 * 256 + 256, device=a.device, dtype=torch.float32) if (prod and shadow is not None) else None
    sup = ctypes.c_int(0)
    nv.check(_L(a).ltx2_gemm_bf16_fold(nv.ptr(a), a.stride(0), nv.ptr(w), nv.ptr(bias), nv.ptr(out), out.stride(0), M, N, K, epilogue, nv.ptr(gate_table),
                                       nv.ptr(shadow), shadow.stride(0) if shadow is not None else 0, nv.ptr(shadow_scale), nv.ptr(ss),
                                       ss.stride(0) if ss is not None else 0,
                                       nv.ptr(rf_parts), rf_parts.stride(0) if rf_parts is not None else 0, rf_parts.shape[0] if rf_parts is not None else 0,
                                       rf_dim, eps, ctypes.byref(sup), nv.stream()))
    if not sup.value:
        return None
    return (out, ss) if prod else out


def flash_attn_rowscale(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, nkv: int, q_ss: torch.Tensor, eps: float = 1e-6,
                        scale: Optional[float] = None) -> torch.Tensor:
    """flash_attn with q's RMS normalisation over its FULL width folded in as a per-row softmax scale (q_ss: partial sums of squares [Nq, P])."""
    assert q.dtype in ACT16 and q_ss.dtype == torch.float32 and q_ss.is_contiguous()
    nq, hd = q.shape[0], vt.shape[1]
    out = torch.empty(nq, heads * hd, device=q.device, dtype=q.dtype)
    if scale is None:
        scale = 1.0 / math.sqrt(float(hd))
    nv.check(_L(q).ltx2_flash_attn_rowscale(nv.ptr(q), q.stride(0), nv.ptr(k), k.stride(0), nv.ptr(vt), vt.shape[2], nv.ptr(out), out.stride(0), nq, nkv,
                                            heads, hd, scale, nv.ptr(q_ss), q_ss.shape[1], heads * hd, eps, nv.stream()))
    return out


def flash_attn_gated(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, nkv: int, gate_logits: torch.Tensor,
                     scale: Optional[float] = None) -> torch.Tensor:
    """flash_attn with per-head output gates 2*sigmoid(gate_logits[q, h]) applied in the epilogue (fp32, before the rounding)."""
    assert q.dtype in ACT16 and gate_logits.dtype == torch.float32 and gate_logits.stride(1) == 1
    nq, hd = q.shape[0], vt.shape[1]
    out = torch.empty(nq, heads * hd, device=q.device, dtype=q.dtype)
    if scale is None:
        scale = 1.0 / math.sqrt(float(hd))
    nv.check(_L(q).ltx2_flash_attn_gated(nv.ptr(q), q.stride(0), nv.ptr(k), k.stride(0), nv.ptr(vt), vt.shape[2], nv.ptr(out), out.stride(0), nq, nkv,
                                         heads, hd, scale, nv.ptr(gate_logits), gate_logits.stride(0), nv.stream()))
    return out


def flash_attn_gated_parts(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, nkv: int, x: torch.Tensor, gate_w: torch.Tensor,
                           gate_b: torch.Tensor, scale: Optional[float] = None) -> torch.Tensor:
    """The engine's many-row form of flash_attn_gated: gate logits x @ gate_w^T as 8 K-slice partial sums, added (with gate_b) in the attention epilogue."""
    assert q.dtype in ACT16 and x.dtype == q.dtype and gate_w.dtype == q.dtype and x.stride(1) == 1
    nq, hd = q.shape[0], vt.shape[1]
    out = torch.empty(nq, heads * hd, device=q.device, dtype=q.dtype)
    parts = torch.empty(8, nq, heads, device=q.device, dtype=torch.float32)
    if scale is None:
        scale = 1.0 / math.sqrt(float(hd))
    nv.check(_L(q).ltx2_flash_attn_gated_parts(nv.ptr(q), q.stride(0), nv.ptr(k), k.stride(0), nv.ptr(vt), vt.shape[2], nv.ptr(out), out.stride(0), nq, nkv,
                                               heads, hd, scale, nv.ptr(x), x.stride(0), nv.ptr(_c(gate_w)), nv.ptr(_c(gate_b.float())), x.shape[1], nv.ptr(parts),
                                               nv.stream()))
    return out


def flash_attn_keymask(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, nkv: int, mask: torch.Tensor,
                       scale: Optional[float] = None) -> torch.Tensor:
    """flash_attn with a key mask: mask [Nkv] bool / 0-1 (True = attend) -- the boolean context mask of the reference's text
    cross-attention (model.py:163-201, attention.py:38-70)."""
    assert q.dtype in ACT16 and mask.numel() == nkv
    nq, hd = q.shape[0], vt.shape[1]
    out = torch.empty(nq, heads * hd, device=q.device, dtype=q.dtype)
    f = (mask.reshape(-1).to(q.device) != 0).to(torch.float32).contiguous()
    words = torch.empty(vt.shape[2] // 64, device=q.device, dtype=torch.int64)
    if scale is None:
        scale = 1.0 / math.sqrt(float(hd))
    nv.check(_L(q).ltx2_flash_attn_keymask(nv.ptr(q), q.stride(0), nv.ptr(k), k.stride(0), nv.ptr(vt), vt.shape[2], nv.ptr(out), out.stride(0), nq, nkv,
                                           heads, hd, scale, nv.ptr(f), nv.ptr(words), nv.stream()))
    return out


def gemm_w8a16(a: torch.Tensor, w8: torch.Tensor, wscale: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = nv.EPI_BF16,
               out: Optional[torch.Tensor] = None, gate_table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ dequant(w8)[N,K]^T + bias) with fp8-RESIDENT weights: w8 = float8_e4m3fn codes (uint8 view
    accepted), wscale fp32 [N]; bit-identical to gemm() on bf16(f32(w8) * wscale[:, None])."""
    assert a.dtype in ACT16 and w8.element_size() == 1 and wscale.dtype == torch.float32 and a.dim() == 2 and w8.dim() == 2
    a, w8, wscale = _c(a), _c(w8), _c(wscale)
    M, K = a.shape
    N = w8.shape[0]
    assert wscale.numel() == N
    if out is None:
        assert epilogue != nv.EPI_RESID_GATE_F32, "RESID_GATE accumulates into `out`; pass it"
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if epilogue == nv.EPI_F32 else a.dtype)
    nv.check(_L(a).ltx2_gemm_w8a16(nv.ptr(a), a.stride(0), nv.ptr(w8), nv.ptr(wscale), nv.ptr(bias), nv.ptr(out), out.stride(0), M, N, K,
                                      epilogue, None, 0, nv.ptr(gate_table), nv.stream()))
    return out


def quantize_rows_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 [R, K] -> (float8_e4m3fn codes as uint8 [R, K], fp32 scale [R]): scale = max|row| / 448 (1 for a zero row),
    code = e4m3fn_rne(x * (1 / scale)).  The quantiser of the fp8 compute path (activations per token, weights per output channel)."""
    assert x.dtype in ACT16 and x.dim() == 2
    x = _c(x)
    R, K = x.shape
    codes = torch.empty(R, K, device=x.device, dtype=torch.uint8)
    scale = torch.empty(R, device=x.device, dtype=torch.float32)
    nv.check(_L(x).ltx2_quantize_rows_fp8(nv.ptr(x), x.stride(0), R, K, nv.ptr(codes), K, nv.ptr(scale), nv.stream()))
    return codes, scale


def gemm_fp8(a8: torch.Tensor, ascale: torch.Tensor, w8: torch.Tensor, wscale: torch.Tensor, bias: Optional[torch.Tensor] = None,
             epilogue: int = nv.EPI_BF16, out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
             gate_table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(ascale[m] * wscale[n] * (f32(a8)[M,K] @ f32(w8)[N,K]^T) + bias) on the fp8 MFMA (a8, w8: float8_e4m3fn codes,
    uint8 views accepted)."""
    assert a8.element_size() == 1 and w8.element_size() == 1 and a8.dim() == 2 and w8.dim() == 2
    assert ascale.dtype == torch.float32 and wscale.dtype == torch.float32
    a8, w8, ascale, wscale = _c(a8), _c(w8), _c(ascale), _c(wscale)
    M, K = a8.shape
    N = w8.shape[0]
    assert ascale.numel() == M and wscale.numel() == N and w8.shape[1] == K
    if out is None:
        assert epilogue != nv.EPI_RESID_GATE_F32, "RESID_GATE accumulates into `out`; pass it"
        out = torch.empty(M, N, device=a8.device, dtype=torch.float32 if epilogue == nv.EPI_F32 else BF16)
    gs = 0
    if gate is not None:
        gate = _c(gate)
        gs = 0 if gate.shape[0] == 1 else gate.stride(0)
    nv.check(nv.lib().ltx2_gemm_fp8(nv.ptr(a8), a8.stride(0), nv.ptr(ascale), nv.ptr(w8), nv.ptr(wscale), nv.ptr(bias), nv.ptr(out), out.stride(0),
                                    M, N, K, epilogue, nv.ptr(gate), gs, nv.ptr(gate_table), nv.stream()))
    return out


def gemm_fp8_qkv_vt(a8, ascale, w8, wscale, bias, heads: int, head_dim: int = 128):
    """gemm_qkv_vt on the fp8 compute path -> (qkv [M, 3D] bf16, vt [H, hd, Npad], fused)."""
    a8, w8, ascale, wscale = _c(a8), _c(w8), _c(ascale), _c(wscale)
    M, K = a8.shape
    N = w8.shape[0]
    D = heads * head_dim
    assert N == 3 * D
    npad = (M + 63) // 64 * 64
    out = torch.empty(M, N, device=a8.device, dtype=BF16)
    vt = torch.empty(heads, head_dim, npad, device=a8.device, dtype=BF16)
    import ctypes
    fused = ctypes.c_int(0)
    nv.check(nv.lib().ltx2_gemm_fp8_qkv_vt(nv.ptr(a8), a8.stride(0), nv.ptr(ascale), nv.ptr(w8), nv.ptr(wscale), nv.ptr(bias), nv.ptr(out),
                                           out.stride(0), M, N, K, nv.ptr(vt), 2 * D, npad, head_dim, ctypes.byref(fused), nv.stream()))
    return out, vt, bool(fused.value)


def gemv(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act_in: int = 0, act_out: int = 0) -> torch.Tensor:
    assert a.dtype == torch.float32 and w.dtype in ACT16
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    nv.check(_L(w).ltx2_gemv_f32(nv.ptr(a), a.stride(0), nv.ptr(w), nv.ptr(bias), nv.ptr(out), N, M, N, K, act_in,
                                    act_out, nv.stream()))
    return out


def conv_weight_to_engine(w: torch.Tensor, d2s_stride: Optional[Tuple[int, int, int]] = None, dtype: torch.dtype = BF16) -> torch.Tensor:
    """PyTorch conv3d weight (Cout, Cin, 3, 3, 3) -> engine layout bf16 [Cout][27][Cin]
    (tap = (kt*3+kh)*3+kw).  For depth-to-space convs the output rows are permuted from
    ch = c*sp + s to n' = s*Cf + c so one contiguous channel run lands on one output voxel."""
    cout, cin = w.shape[0], w.shape[1]
    e = w.permute(0, 2, 3, 4, 1).reshape(cout, 27, cin)
    if d2s_stride is not None:
        sp = d2s_stride[0] * d2s_stride[1] * d2s_stride[2]
        cf = cout // sp
        e = e.reshape(cf, sp, 27, cin).permute(1, 0, 2, 3).reshape(cout, 27, cin)
    return e.to(dtype).contiguous()


def conv_bias_to_engine(b: torch.Tensor, d2s_stride: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
    if d2s_stride is not None:
        sp = d2s_stride[0] * d2s_stride[1] * d2s_stride[2]
        b = b.reshape(-1, sp).t().reshape(-1)
    return b.float().contiguous()


def conv3d(x: torch.Tensor, w_engine: torch.Tensor, bias: Optional[torch.Tensor], causal: bool = False, mode: int = 0,
           res: Optional[torch.Tensor] = None, stride: Tuple[int, int, int] = (1, 1, 1), residual: bool = False,
           pad_zero: int = 0) -> torch.Tensor:
    """x bf16 [T,H,W,Cin] channels-last; w_engine [Cout][27 or 9][Cin] from conv_weight_to_engine /
    conv2d_weight_to_engine (9 taps = per-frame 3x3 conv).  pad_zero: 0 reflect H/W + replicate T (VAE decoder),
    1 / True zero padding in T/H/W (spatial upscaler), 2 zero padding in H/W + replicate T (VAE encoder)."""
    assert x.dtype in ACT16 and x.dim() == 4 and w_engine.dtype == x.dtype
    x = _c(x)
    T, H, W, Cin = x.shape
    Cout = w_engine.shape[0]
    kt = w_engine.shape[1] // 9
    ft, fh, fw = stride
    if mode == 2:
        sp = ft * fh * fw
        out = torch.empty(T * ft - (1 if ft > 1 else 0), H * fh, W * fw, Cout // sp, device=x.device, dtype=x.dtype)
    else:
        out = torch.empty(T, H, W, Cout, device=x.device, dtype=x.dtype)
    nv.check(_L(x).ltx2_conv3d_fused(nv.ptr(x), nv.ptr(w_engine), nv.ptr(bias), nv.ptr(out), T, H, W, Cin, Cout,
                                        int(causal), mode, nv.ptr(res), ft, fh, fw, int(residual), int(pad_zero), kt,
                                        nv.stream()))
    return out


def conv2d_weight_to_engine(w: torch.Tensor, pixel_shuffle: int = 0, dtype: torch.dtype = BF16) -> torch.Tensor:
    """PyTorch conv2d weight (Cout, Cin, 3, 3) -> engine layout bf16 [Cout][9][Cin] (tap = kh*3+kw).  With
    pixel_shuffle = r the output rows are permuted from ch = c*r*r + s (PyTorch pixel_shuffle packing
    (C, r_h, r_w)) to n' = s*Cf + c for the depth-to-space epilogue."""
    cout, cin = w.shape[0], w.shape[1]
    e = w.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    if pixel_shuffle:
        sp = pixel_shuffle * pixel_shuffle
        e = e.reshape(cout // sp, sp, 9, cin).permute(1, 0, 2, 3).reshape(cout, 9, cin)
    return e.to(dtype).contiguous()


def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5,
                   res: Optional[torch.Tensor] = None, act: bool = True) -> torch.Tensor:
    """y = [silu](GroupNorm(x over (C/groups, T, H, W)) * gamma + beta + res) on channels-last bf16 [..., C]."""
    assert x.dtype in ACT16 and x.is_contiguous()
    C = x.shape[-1]
    P = x.numel() // C
    y = torch.empty_like(x)
    sums = torch.empty(2 * groups * (1 + (P + 15) // 16), device=x.device, dtype=torch.float32)
    nv.check(_L(x).ltx2_groupnorm_silu(nv.ptr(x), nv.ptr(res), nv.ptr(y), P, C, groups, eps, nv.ptr(_c(gamma.float())),
                                          nv.ptr(_c(beta.float())), nv.ptr(sums), int(act), nv.stream()))
    return y


def s2d_downsample(y: torch.Tensor, x: torch.Tensor, stride: Tuple[int, int, int]) -> torch.Tensor:
    """space_to_depth(y) + group_mean(space_to_depth(x)) on channels-last bf16 [T,H,W,C] (VAE encoder downsample)."""
    assert y.dtype in ACT16 and x.dtype == y.dtype and y.shape[:3] == x.shape[:3]
    y, x = _c(y), _c(x)
    T, H, W, Cc = y.shape
    st, sh, sw = stride
    out = torch.empty(T // st, H // sh, W // sw, Cc * st * sh * sw, device=y.device, dtype=y.dtype)
    nv.check(_L(y).ltx2_s2d_downsample(nv.ptr(y), nv.ptr(x), nv.ptr(out), T, H, W, Cc, x.shape[3], st, sh, sw, nv.stream()))
    return out


def latent_unnormalize_nhwc(latent: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, dtype: torch.dtype = BF16) -> torch.Tensor:
    """latent fp32 [C,T,H,W] -> bf16 [T,H,W,C] = latent * std[c] + mean[c]  (PerChannelStatistics.un_normalize,
    video_vae/ops.py:158-171; same kernel as the VAE decoder's input stage)."""
    assert latent.dtype == torch.float32 and latent.dim() == 4
    latent = _c(latent)
    C, T, H, W = latent.shape
    out = torch.empty(T, H, W, C, device=latent.device, dtype=dtype)
    nv.check(nv.lib(dtype).ltx2_vae_prepare_latent(nv.ptr(latent), nv.ptr(_c(std.float())), nv.ptr(_c(mean.float())), None, 0.0,
                                              nv.ptr(out), C, T * H * W, nv.stream()))
    return out


def latent_normalize_nchw(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """x bf16 [T,H,W,C] -> fp32 [C,T,H,W] = (x - mean[c]) / std[c]."""
    assert x.dtype in ACT16 and x.dim() == 4 and x.is_contiguous()
    T, H, W, C = x.shape
    out = torch.empty(C, T, H, W, device=x.device, dtype=torch.float32)
    nv.check(_L(x).ltx2_latent_normalize_nchw(nv.ptr(x), nv.ptr(_c(mean.float())), nv.ptr(_c(std.float())), nv.ptr(out), C,
                                                 T * H * W, nv.stream()))
    return out


def adaln_rmsnorm(x: torch.Tensor, eps: float = 1e-6, layer_norm: bool = False,
                  scale_tab: Optional[torch.Tensor] = None, shift_tab: Optional[torch.Tensor] = None,
                  scale_emb: Optional[torch.Tensor] = None, shift_emb: Optional[torch.Tensor] = None,
                  emb_stride: int = 0, dtype: torch.dtype = BF16) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.dim() == 2
    x = _c(x)
    rows, D = x.shape
    out = torch.empty(rows, D, device=x.device, dtype=dtype)
    nv.check(nv.lib(dtype).ltx2_adaln_rmsnorm(nv.ptr(x), D, nv.ptr(out), D, rows, D, eps, int(layer_norm), nv.ptr(scale_tab),
                                         nv.ptr(shift_tab), nv.ptr(scale_emb), nv.ptr(shift_emb), emb_stride, nv.stream()))
    return out


def adaln_rmsnorm_fp8(x: torch.Tensor, eps: float = 1e-6, layer_norm: bool = False, scale_tab: Optional[torch.Tensor] = None,
                      shift_tab: Optional[torch.Tensor] = None, scale_emb: Optional[torch.Tensor] = None,
                      shift_emb: Optional[torch.Tensor] = None, emb_stride: int = 0, want_bf16: bool = True):
    """adaln_rmsnorm with the per-token e4m3fn quantiser fused in -> (bf16 out or None, codes uint8 [rows, D], scale fp32 [rows])."""
    assert x.dtype == torch.float32 and x.dim() == 2
    x = _c(x)
    rows, D = x.shape
    out = torch.empty(rows, D, device=x.device, dtype=BF16) if want_bf16 else None
    codes = torch.empty(rows, D, device=x.device, dtype=torch.uint8)
    scale = torch.empty(rows, device=x.device, dtype=torch.float32)
    nv.check(nv.lib().ltx2_adaln_rmsnorm_fp8(nv.ptr(x), D, nv.ptr(out), D, nv.ptr(codes), D, nv.ptr(scale), rows, D, eps, int(layer_norm),
                                             nv.ptr(scale_tab), nv.ptr(shift_tab), nv.ptr(scale_emb), nv.ptr(shift_emb), emb_stride, nv.stream()))
    return out, codes, scale


def qknorm_rope_(buf: torch.Tensor, D: int, head_dim: int, q_off: int, q_weight: torch.Tensor,
                 k_off: int = 0, k_weight: Optional[torch.Tensor] = None, eps: float = 1e-6,
                 cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert buf.dtype in ACT16 and buf.dim() == 2 and buf.is_contiguous()
    nv.check(_L(buf).ltx2_qknorm_rope(nv.ptr(buf), buf.stride(0), buf.shape[0], D, head_dim, q_off, nv.ptr(q_weight),
                                       k_off, nv.ptr(k_weight), eps, nv.ptr(cos), nv.ptr(sin), nv.stream()))
    return buf


def vt_transpose(v: torch.Tensor, heads: int, head_dim: int = 128) -> torch.Tensor:
    """v bf16 [Nkv, >= heads*head_dim] (a strided column view is fine) -> VT [H,head_dim,Npad]."""
    assert v.dtype in ACT16 and v.dim() == 2 and v.stride(1) == 1
    nkv = v.shape[0]
    npad = (nkv + 63) // 64 * 64
    vt = torch.empty(heads, head_dim, npad, device=v.device, dtype=v.dtype)
    nv.check(_L(v).ltx2_vt_transpose(nv.ptr(v), v.stride(0), nv.ptr(vt), nkv, npad, heads, head_dim, nv.stream()))
    return vt


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, nkv: int,
               scale: Optional[float] = None) -> torch.Tensor:
    """q [Nq, H*hd], k [Nkv, H*hd] bf16 (row-strided views allowed), vt [H,hd,Npad] from vt_transpose (hd 128 or 64)."""
    assert q.dtype in ACT16 and k.dtype == q.dtype and vt.dtype == q.dtype and q.stride(1) == 1 and k.stride(1) == 1
    nq, hd = q.shape[0], vt.shape[1]
    out = torch.empty(nq, heads * hd, device=q.device, dtype=q.dtype)
    if scale is None:
        scale = 1.0 / math.sqrt(float(hd))
    nv.check(_L(q).ltx2_flash_attn(nv.ptr(q), q.stride(0), nv.ptr(k), k.stride(0), nv.ptr(vt), vt.shape[2], nv.ptr(out),
                                      out.stride(0), nq, nkv, heads, hd, scale, nv.stream()))
    return out


def attn_head_gate_(att: torch.Tensor, x: torch.Tensor, gate_w: torch.Tensor, gate_b: torch.Tensor, heads: int) -> torch.Tensor:
    """In place: att [rows, H*hd] bf16 *= 2*sigmoid(x @ gate_w^T + gate_b) per head; returns the fp32 logits."""
    assert att.dtype in ACT16 and x.dtype == att.dtype and gate_w.dtype == att.dtype and att.is_contiguous() and x.stride(1) == 1
    rows, hd = att.shape[0], att.shape[1] // heads
    logits = torch.empty(rows, heads, device=att.device, dtype=torch.float32)
    nv.check(_L(att).ltx2_attn_head_gate(nv.ptr(att), att.stride(0), nv.ptr(x), x.stride(0), nv.ptr(_c(gate_w)), nv.ptr(_c(gate_b.float())),
                                          nv.ptr(logits), rows, x.shape[1], heads, hd, nv.stream()))
    return logits


def rope_tables(positions: torch.Tensor, dim: int, theta: float, max_pos) -> Tuple[torch.Tensor, torch.Tensor]:
    """positions (1, n_dims, N, 2) on the GPU -> SPLIT-RoPE cos, sin fp32 [N, dim/2] (slot h*(d/2)+j for head h)."""
    pos = _c(positions[0].float())
    n_dims, N = pos.shape[0], pos.shape[1]
    if n_dims != len(max_pos):
        raise ValueError(f"Number of position dimensions ({n_dims}) must match max_pos length ({len(max_pos)})")
    n_freq = dim // (2 * n_dims)
    grid = (torch.tensor(float(theta)) ** torch.linspace(0.0, 1.0, n_freq, dtype=torch.float32) * (math.pi / 2)).float().to(pos.device)
    mp = torch.tensor([float(m) for m in max_pos], device=pos.device)
    cos = torch.empty(N, dim // 2, device=pos.device, dtype=torch.float32)
    sin = torch.empty_like(cos)
    nv.check(nv.lib().ltx2_rope_tables(nv.ptr(pos), nv.ptr(grid), nv.ptr(mp), N, n_dims, n_freq, dim // 2, nv.ptr(cos), nv.ptr(sin),
                                       nv.stream()))
    return cos, sin


def timestep_sinusoid(t: torch.Tensor, mult: float, dim: int = 256) -> torch.Tensor:
    t = _c(t.float())
    out = torch.empty(t.numel(), dim, device=t.device, dtype=torch.float32)
    nv.check(nv.lib().ltx2_timestep_sinusoid(nv.ptr(t), 1, mult, t.numel(), dim, nv.ptr(out), None, nv.stream()))
    return out


def dequant_fp8(w: torch.Tensor, scale: float, dtype: torch.dtype = BF16) -> torch.Tensor:
    """fp8 e4m3fn weight (device tensor, torch.float8_e4m3fn or its uint8 view) * scale -> bf16."""
    raw = w.view(torch.uint8) if w.dtype != torch.uint8 else w
    raw = _c(raw)
    out = torch.empty(raw.shape, device=raw.device, dtype=dtype)
    nv.check(nv.lib(dtype).ltx2_dequant_fp8_e4m3fn(nv.ptr(raw), float(scale), nv.ptr(out), raw.numel(), nv.stream()))
    return out


def cast_bf16(x: torch.Tensor, dtype: torch.dtype = BF16) -> torch.Tensor:
    x = _c(x.float())
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    nv.check(nv.lib(dtype).ltx2_cast_f32_bf16(nv.ptr(x), nv.ptr(out), x.numel(), nv.stream()))
    return out


def x0_from_velocity(latent: torch.Tensor, velocity: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    """latent/velocity fp32 [N,C]; timesteps fp32 with 1 or N elements."""
    latent, velocity, timesteps = _c(latent), _c(velocity), _c(timesteps.float())
    n, c = latent.shape
    out = torch.empty_like(latent)
    stride = 0 if timesteps.numel() == 1 else 1
    nv.check(nv.lib().ltx2_x0_from_velocity(nv.ptr(latent), nv.ptr(velocity), nv.ptr(timesteps), stride, 0.0, nv.ptr(out),
                                            n, c, nv.stream()))
    return out


def euler_step(x: torch.Tensor, x0: torch.Tensor, sigma: float, sigma_next: float,
               mask: Optional[torch.Tensor] = None, clean: Optional[torch.Tensor] = None) -> torch.Tensor:
    x, x0 = _c(x), _c(x0)
    n, c = x.shape
    out = torch.empty_like(x)
    nv.check(nv.lib().ltx2_euler_step(nv.ptr(x), nv.ptr(x0), nv.ptr(mask), nv.ptr(clean), float(sigma), float(sigma_next),
                                      nv.ptr(out), n, c, nv.stream()))
    return out


def pixnorm_mod_silu(x: torch.Tensor, table: torch.Tensor, te: Optional[torch.Tensor], shift_row: int, scale_row: int,
                     eps: float = 1e-6) -> torch.Tensor:
    assert x.dtype in ACT16
    x = _c(x)
    C_ = x.shape[-1]
    P = x.numel() // C_
    y = torch.empty_like(x)
    nv.check(_L(x).ltx2_pixnorm_mod_silu(nv.ptr(x), nv.ptr(y), P, C_, eps, nv.ptr(table), nv.ptr(te), shift_row,
                                            scale_row, nv.stream()))
    return y


def video_chunk_to_uint8(cur: torch.Tensor, frames: torch.Tensor, t_dst0: int, prev: Optional[torch.Tensor] = None,
                         ramp: Optional[torch.Tensor] = None) -> None:
    """One temporal chunk `cur` fp32 [3,Tc,H,W] -> frames uint8 [T,H,W,3] at frame t_dst0, cross-faded with the tail of the previous
    chunk `prev` over len(ramp) frames, trimmed at T (reference simple_decoder.py:760-798) -- one pass instead of cat / blend / cat /
    convert over fp32 volumes."""
    cur = _c(cur.float())
    _, Tc, H, W = cur.shape
    ov = 0 if prev is None else int(ramp.numel())
    if prev is not None:
        prev, ramp = _c(prev.float()), _c(ramp.float())
    nv.check(nv.lib().ltx2_video_chunk_to_uint8(nv.ptr(cur), nv.ptr(prev), nv.ptr(ramp), nv.ptr(frames), Tc, 0 if prev is None else prev.shape[1], ov,
                                                H, W, int(t_dst0), frames.shape[0], nv.stream()))


def tile_blend_accumulate(tile: torch.Tensor, nt: int, nh: int, nw: int, mt: torch.Tensor, mh: torch.Tensor, mw: torch.Tensor,
                          out: torch.Tensor, wsum: torch.Tensor, t0: int, h0: int, w0: int) -> None:
    """out[3,OT,OH,OW] += tile[3,dt,dh,dw][:, :nt, :nh, :nw] * (mt x mh x mw); wsum[OT,OH,OW] += mask (reference tiling.py:380-404)."""
    tile = _c(tile.float())
    _, dt, dh, dw = tile.shape
    _, OT, OH, OW = out.shape
    nv.check(nv.lib().ltx2_tile_blend_accumulate(nv.ptr(tile), dt, dh, dw, nt, nh, nw, nv.ptr(_c(mt.float())), nv.ptr(_c(mh.float())),
                                                 nv.ptr(_c(mw.float())), nv.ptr(out), nv.ptr(wsum), OT, OH, OW, t0, h0, w0, nv.stream()))


def tile_blend_finish(out: torch.Tensor, wsum: torch.Tensor) -> None:
    """out /= clamp(wsum, 1e-8) in place (reference tiling.py:410-412)."""
    nv.check(nv.lib().ltx2_tile_blend_finish(nv.ptr(out), nv.ptr(wsum), wsum.numel(), nv.stream()))


def video_to_uint8(video: torch.Tensor) -> torch.Tensor:
    """video fp32 [3,T,H,W] -> uint8 [T,H,W,3] (reference simple_decoder.py:792-798)."""
    video = _c(video.float())
    _, T, H, W = video.shape
    out = torch.empty(T, H, W, 3, device=video.device, dtype=torch.uint8)
    nv.check(nv.lib().ltx2_video_to_uint8(nv.ptr(video), nv.ptr(out), T, H, W, nv.stream()))
    return out
