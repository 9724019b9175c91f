"""Triangulation of the leaf arithmetic (parity item 4d): the golden vectors pin the reference's COMPOSITION through
tools/mlx_shim.py, and both the shim and the oracle state MLX's leaf ops with torch calls -- a shared misreading of one of them
would pass every other test.  Here each leaf op is written a THIRD time, in float64 numpy, straight from MLX's documented
definitions (mlx.core.fast.rms_norm / scaled_dot_product_attention, mlx.core.conv2d on NHWC input with (C_out, kH, kW, C_in)
weights, mlx.nn.gelu_approx, mlx.nn.LayerNorm, mlx.nn.silu, mlx.nn.Linear) with explicit loops / einsums and no torch, and
compared with BOTH the shim's and the oracle's statements on random inputs."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RNG = np.random.RandomState(7)


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    assert err < tol, err


# ---- float64 numpy restatements of MLX's documented definitions -------------------------------------------------------------
def np_rms_norm(x, w, eps):
    """mx.fast.rms_norm: x * rsqrt(mean(x^2, last axis) + eps) [* weight]; eps INSIDE the square root."""
    y = x / np.sqrt((x * x).mean(-1, keepdims=True) + eps)
    return y if w is None else y * w


def np_sdpa(q, k, v, scale):
    """mx.fast.scaled_dot_product_attention on (B, heads, T, d): softmax(scale * q k^T, over keys) v."""
    out = np.zeros(q.shape[:-1] + (v.shape[-1],))
    for b in range(q.shape[0]):
        for h in range(q.shape[1]):
            s = scale * (q[b, h] @ k[b, h].T)
            p = np.exp(s - s.max(-1, keepdims=True))
            out[b, h] = (p / p.sum(-1, keepdims=True)) @ v[b, h]
    return out


def np_conv2d_nhwc(x, w, pad):
    """mx.conv2d: input (N, H, W, C_in), weight (C_out, kH, kW, C_in), cross-correlation, symmetric ZERO padding, stride 1."""
    n, h, wd, ci = x.shape
    co, kh, kw, _ = w.shape
    xp = np.zeros((n, h + 2 * pad, wd + 2 * pad, ci))
    xp[:, pad:pad + h, pad:pad + wd] = x
    out = np.zeros((n, h + 2 * pad - kh + 1, wd + 2 * pad - kw + 1, co))
    for a in range(kh):
        for b in range(kw):
            out += np.einsum("nhwc,oc->nhwo", xp[:, a:a + out.shape[1], b:b + out.shape[2]], w[:, a, b])
    return out


def np_gelu_approx(x):
    """mlx.nn.gelu_approx: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))."""
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def np_layer_norm(x, eps, weight=None, bias=None):
    """mlx.nn.LayerNorm: (x - E[x]) / sqrt(Var[x] + eps) (biased variance over the last axis) [* weight + bias]."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps)
    return y if weight is None else y * weight + bias


def np_silu(x):
    return x / (1.0 + np.exp(-x))


def np_conv3d_reflect_replicate(x, w, b, causal):
    """Conv3dSimple (reference simple_decoder.py:90-180) from its description: channels-first (B, C, T, H, W), 3x3x3 stride 1,
    reflect-pad H and W by 1 (edge NOT repeated), replicate-pad T (1 + 1, or 2 leading frames when causal), + bias."""
    bsz, ci, t, h, wd = x.shape
    co = w.shape[0]
    hh = [1] + list(range(h)) + [h - 2]
    ww = [1] + list(range(wd)) + [wd - 2]
    tt = ([0, 0] + list(range(t))) if causal else ([0] + list(range(t)) + [t - 1])
    xp = x[:, :, tt][:, :, :, hh][:, :, :, :, ww]
    out = np.zeros((bsz, co, t, h, wd))
    for a in range(3):
        for bb in range(3):
            for c in range(3):
                out += np.einsum("bcthw,oc->bothw", xp[:, :, a:a + t, bb:bb + h, c:c + wd], w[:, :, a, bb, c])
    return out + b[None, :, None, None, None]


# ---- the shim's and the oracle's statements against them ----------------------------------------------------------------------
def shim():
    from tools import mlx_shim
    return mlx_shim


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def test_rms_norm_three_ways():
    from oracle import dit
    x, w = RNG.randn(2, 5, 96), 1.0 + 0.1 * RNG.randn(96)
    for eps in (1e-6, 1e-5):
        ref = np_rms_norm(x, w, eps)
        close(shim()._Fast.rms_norm(shim().Arr(T(x)), shim().Arr(T(w)), eps).t.numpy(), ref)
        close(dit.rms_norm(T(x), T(w), eps).numpy(), ref)
        close(dit.rms_norm(T(x), None, eps).numpy(), np_rms_norm(x, None, eps))
    # eps placement matters at this scale: rsqrt(mean + eps) != 1 / (sqrt(mean) + eps)
    tiny = 1e-3 * x
    wrong = tiny / (np.sqrt((tiny * tiny).mean(-1, keepdims=True)) + 1e-5) * w
    assert np.abs(wrong - np_rms_norm(tiny, w, 1e-5)).max() > 1e-3


def test_sdpa_three_ways():
    from oracle import dit
    b, h, tq, tk, d = 1, 3, 7, 11, 16
    q, k, v = RNG.randn(b, h, tq, d), RNG.randn(b, h, tk, d), RNG.randn(b, h, tk, d)
    ref = np_sdpa(q, k, v, 1.0 / math.sqrt(d))
    A = shim().Arr
    close(shim()._Fast.scaled_dot_product_attention(A(T(q)), A(T(k)), A(T(v)), scale=1.0 / math.sqrt(d)).t.numpy(), ref)
    # the oracle's sdpa takes tokens-major (B, T, heads*d) and returns the same layout
    merge = lambda a: T(a).permute(0, 2, 1, 3).reshape(b, a.shape[2], h * d)
    close(dit.sdpa(merge(q), merge(k), merge(v), h).numpy(), np.transpose(ref, (0, 2, 1, 3)).reshape(b, tq, h * d))


def test_conv2d_nhwc_and_conv3d_three_ways():
    from oracle import vae
    x, w = RNG.randn(2, 6, 5, 4), RNG.randn(8, 3, 3, 4)
    close(shim().conv2d(shim().Arr(T(x)), shim().Arr(T(w)), padding=1).t.numpy(), np_conv2d_nhwc(x, w, 1))
    close(shim().conv2d(shim().Arr(T(x)), shim().Arr(T(w)), padding=0).t.numpy(), np_conv2d_nhwc(x, w, 0))
    x5, w5, b5 = RNG.randn(1, 4, 3, 5, 6), RNG.randn(6, 4, 3, 3, 3), RNG.randn(6)
    for causal in (False, True):
        close(vae.conv3d_simple(T(x5), T(w5), T(b5), causal).numpy(), np_conv3d_reflect_replicate(x5, w5, b5, causal))
    # the reference composes its 3-D conv from three mx.conv2d calls (one per temporal tap, simple_decoder.py:136-178): the same sum
    xpad = np.concatenate([x5[:, :, :1], x5, x5[:, :, -1:]], axis=2)
    hh, ww = [1] + list(range(5)) + [3], [1] + list(range(6)) + [4]
    xpad = xpad[:, :, :, hh][:, :, :, :, ww]
    acc = np.zeros((1, 3, 5, 6, 6))
    for kt in range(3):
        frames = np.transpose(xpad[0, :, kt:kt + 3], (1, 2, 3, 0))                      # (T, H+2, W+2, Cin) = NHWC batch of frames
        acc[0] += np_conv2d_nhwc(frames, np.transpose(w5[:, :, kt], (0, 2, 3, 1)), 0)
    close(np.transpose(acc, (0, 4, 1, 2, 3)) + b5[None, :, None, None, None], np_conv3d_reflect_replicate(x5, w5, b5, False), tol=1e-12)


def test_activations_layernorm_linear_three_ways():
    import torch.nn.functional as F
    from oracle import dit
    x = 3.0 * RNG.randn(4, 9, 32)
    A = shim().Arr
    close(shim().gelu_approx(A(T(x))).t.numpy(), np_gelu_approx(x))
    close(F.gelu(T(x), approximate="tanh").numpy(), np_gelu_approx(x))                  # what oracle.dit.feed_forward calls
    close(shim().silu(A(T(x))).t.numpy(), np_silu(x))
    close(F.silu(T(x)).numpy(), np_silu(x))
    ln = shim().LayerNorm(32, eps=1e-6, affine=False)
    close(ln(A(T(x))).t.numpy(), np_layer_norm(x, 1e-6))
    close(F.layer_norm(T(x), (32,), eps=1e-6).numpy(), np_layer_norm(x, 1e-6))          # oracle.dit.velocity_model's output norm
    w, b = RNG.randn(12, 32), RNG.randn(12)
    lin = shim().Linear(32, 12)
    lin.weight, lin.bias = A(T(w)), A(T(b))
    close(lin(A(T(x))).t.numpy(), x @ w.T + b)                                          # mlx.nn.Linear: weight is (out, in)
    close(dit.linear(T(x), {"p.weight": T(w), "p.bias": T(b)}, "p").numpy(), x @ w.T + b)
    # the exact erf GELU is NOT what the reference uses (feed_forward.py:26 nn.gelu_approx): the two differ measurably
    assert np.abs(F.gelu(T(x)).numpy() - np_gelu_approx(x)).max() > 1e-4
