"""Throw-away `mlx` -> torch shim, used ONLY by tools/pin_oracle_against_reference.py inside the
build container to execute the reference's own Python files (which import mlx at module top) and
record golden vectors.  It restates MLX *leaf* ops (matmul, rms_norm, SDPA, conv2d, ...) with
their textbook definitions on CPU fp32 torch tensors, so what it pins is the reference's
composition: layouts, orders, slicing, constants.  Never shipped to the GPU box, never imported by
the product or the tests."""
from __future__ import annotations

import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F


def _unwrap(x):
    if isinstance(x, Arr):
        return x.t
    if isinstance(x, (list, tuple)):
        return type(x)(_unwrap(v) for v in x)
    return x


def _wrap(x):
    if isinstance(x, torch.Tensor):
        return Arr(x)
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


class Dtype:
    def __init__(self, td, name):
        self.td, self.name = td, name

    def __repr__(self):
        return f"mlx.core.{self.name}"

    def __eq__(self, o):
        return isinstance(o, Dtype) and o.td == self.td

    def __hash__(self):
        return hash(self.td)


float32, float16, bfloat16 = Dtype(torch.float32, "float32"), Dtype(torch.float16, "float16"), Dtype(torch.bfloat16, "bfloat16")
int32, int64, uint8, bool_ = Dtype(torch.int32, "int32"), Dtype(torch.int64, "int64"), Dtype(torch.uint8, "uint8"), Dtype(torch.bool, "bool_")
_DT = {d.td: d for d in (float32, float16, bfloat16, int32, int64, uint8, bool_)}
_DT[torch.float64] = float32


def _td(d):
    return d.td if isinstance(d, Dtype) else d


def _fix_index(idx):
    """numpy-style index -> (torch index, dims to flip) supporting negative-step slices."""
    if not isinstance(idx, tuple):
        idx = (idx,)
    out, flips, dim = [], [], 0
    for it in idx:
        if it is None:
            out.append(None)
            if dim is not None:
                dim += 1
            continue
        if it is Ellipsis:
            out.append(it)
            dim = None
            continue
        if isinstance(it, slice) and it.step is not None and it.step < 0:
            assert it.step == -1 and it.start is None and it.stop is None and dim is not None, "shim: only [::-1]"
            out.append(slice(None))
            flips.append(dim)
            dim += 1
            continue
        out.append(_unwrap(it))
        if dim is not None and not isinstance(it, int):
            dim += 1
    return tuple(out), flips


class Arr:
    __array_priority__ = 1000

    def __init__(self, v, dtype=None):
        # doubles as `mx.array(...)`: accepts tensors, Arr, numpy, python scalars / lists
        if isinstance(v, torch.Tensor):
            t = v
        elif isinstance(v, Arr):
            t = v.t
        elif isinstance(v, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(v))
        else:
            t = torch.tensor(_unwrap(v))
        if dtype is not None:
            t = t.to(_td(dtype))
        elif t.dtype == torch.float64:
            t = t.float()
        self.t = t

    # --- attributes ---
    shape = property(lambda s: tuple(s.t.shape))
    ndim = property(lambda s: s.t.ndim)
    size = property(lambda s: s.t.numel())
    dtype = property(lambda s: _DT[s.t.dtype])
    T = property(lambda s: Arr(s.t.T))

    def astype(self, d):
        td = _td(d)
        if td == torch.uint8:
            return Arr(self.t.to(torch.float32).to(torch.uint8) if self.t.is_floating_point() else self.t.to(td))
        return Arr(self.t.to(td))

    def reshape(self, *s):
        if len(s) == 1 and isinstance(s[0], (tuple, list)):
            s = tuple(s[0])
        return Arr(self.t.reshape(*s))

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes:
            axes = tuple(reversed(range(self.t.ndim)))
        return Arr(self.t.permute(*axes))

    def flatten(self, start=0, end=-1):
        return Arr(self.t.flatten(start, end))

    def squeeze(self, axis=None):
        return Arr(self.t.squeeze() if axis is None else self.t.squeeze(axis))

    def item(self):
        return self.t.item()

    def tolist(self):
        return self.t.tolist()

    def sum(self, axis=None, keepdims=False):
        return Arr(self.t.sum() if axis is None else self.t.sum(dim=axis, keepdim=keepdims))

    def mean(self, axis=None, keepdims=False):
        return Arr(self.t.mean() if axis is None else self.t.mean(dim=axis, keepdim=keepdims))

    def var(self, axis=None, keepdims=False, ddof=0):        # mx.array.var: population variance by default
        return Arr(self.t.var(unbiased=bool(ddof)) if axis is None else self.t.var(dim=axis, keepdim=keepdims, unbiased=bool(ddof)))

    def max(self, axis=None, keepdims=False):
        return Arr(self.t.max() if axis is None else self.t.amax(dim=axis, keepdim=keepdims))

    def min(self, axis=None, keepdims=False):
        return Arr(self.t.min() if axis is None else self.t.amin(dim=axis, keepdim=keepdims))

    def __array__(self, dtype=None, copy=None):
        a = self.t.detach().float().numpy() if self.t.dtype == torch.bfloat16 else self.t.detach().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __float__(self):
        return float(self.t)

    def __int__(self):
        return int(self.t)

    def __bool__(self):
        return bool(self.t)

    def __len__(self):
        return self.t.shape[0]

    def __iter__(self):
        return (Arr(v) for v in self.t)

    def __getitem__(self, idx):
        ti, flips = _fix_index(idx)
        r = self.t[ti]
        return Arr(r.flip(flips) if flips else r)

    def __repr__(self):
        return f"array({self.t})"


def _bin(name, rname=None):
    def f(a, b):
        return Arr(getattr(torch.Tensor, name)(_t(a, b), _t(b, a)))

    def r(a, b):
        return Arr(getattr(torch.Tensor, name)(_t(b, a), _t(a, b)))

    setattr(Arr, f"__{name.strip('_')}__", f)
    if rname:
        setattr(Arr, rname, r)


def _t(x, like=None):
    x = _unwrap(x)
    if isinstance(x, torch.Tensor):
        return x
    ref = _unwrap(like)
    if isinstance(ref, torch.Tensor) and ref.is_floating_point():
        return torch.tensor(x, dtype=ref.dtype)
    return torch.tensor(x)


for _n, _r in (("__add__", "__radd__"), ("__sub__", "__rsub__"), ("__mul__", "__rmul__"), ("__truediv__", "__rtruediv__"),
               ("__pow__", "__rpow__"), ("__matmul__", "__rmatmul__"), ("__floordiv__", "__rfloordiv__")):
    def _mk(n):
        def f(a, b):
            return Arr(getattr(torch.Tensor, n)(_t(a, b), _t(b, a)))

        def r(a, b):
            return Arr(getattr(torch.Tensor, n)(_t(b, a), _t(a, b)))
        return f, r
    _f, _rf = _mk(_n)
    setattr(Arr, _n, _f)
    setattr(Arr, _r, _rf)
for _n in ("__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__"):
    def _mkc(n):
        return lambda a, b: Arr(getattr(torch.Tensor, n)(_t(a, b), _t(b, a)))
    setattr(Arr, _n, _mkc(_n))
Arr.__neg__ = lambda a: Arr(-a.t)
Arr.__hash__ = lambda a: id(a)


# ------------------------------------------------------------------ mlx.core
array = Arr


def _shape(s):
    return tuple(s) if isinstance(s, (tuple, list)) else (s,)


def zeros(shape, dtype=float32):
    return Arr(torch.zeros(_shape(shape), dtype=_td(dtype)))


def ones(shape, dtype=float32):
    return Arr(torch.ones(_shape(shape), dtype=_td(dtype)))


def full(shape, v, dtype=float32):
    return Arr(torch.full(_shape(shape), float(_unwrap(v)) if not isinstance(_unwrap(v), torch.Tensor) else _unwrap(v).item(), dtype=_td(dtype)))


zeros_like = lambda a: Arr(torch.zeros_like(a.t))
ones_like = lambda a: Arr(torch.ones_like(a.t))


def arange(*a, dtype=None):
    a = [_unwrap(x) for x in a]
    isf = any(isinstance(x, float) for x in a)
    t = torch.arange(*a, dtype=_td(dtype) if dtype is not None else (torch.float32 if isf else torch.int32))
    return Arr(t)


def linspace(a, b, n=50, dtype=float32):
    return Arr(torch.linspace(float(a), float(b), int(n), dtype=_td(dtype)))


concatenate = lambda xs, axis=0: Arr(torch.cat([_t(x) for x in xs], dim=axis))
stack = lambda xs, axis=0: Arr(torch.stack([_t(x) for x in xs], dim=axis))


def meshgrid(*xs, indexing="xy"):
    return tuple(Arr(g) for g in torch.meshgrid(*[_t(x) for x in xs], indexing=indexing))


cos = lambda a: Arr(torch.cos(_t(a)))
sin = lambda a: Arr(torch.sin(_t(a)))
exp = lambda a: Arr(torch.exp(_t(a)))
log = lambda a: Arr(torch.log(_t(a)))
sqrt = lambda a: Arr(torch.sqrt(_t(a)))
rsqrt = lambda a: Arr(torch.rsqrt(_t(a)))
tanh = lambda a: Arr(torch.tanh(_t(a)))
sigmoid = lambda a: Arr(torch.sigmoid(_t(a)))
abs = lambda a: Arr(torch.abs(_t(a)))  # noqa: A001
square = lambda a: Arr(_t(a) ** 2)
power = lambda a, b: Arr(torch.pow(_t(a, b), _t(b, a)))
maximum = lambda a, b: Arr(torch.maximum(_t(a, b), _t(b, a)))
minimum = lambda a, b: Arr(torch.minimum(_t(a, b), _t(b, a)))
where = lambda c, a, b: Arr(torch.where(_t(c), _t(a, b), _t(b, a)))
clip = lambda a, lo, hi: Arr(torch.clamp(_t(a), lo, hi))
repeat = lambda a, n, axis=None: Arr(torch.repeat_interleave(_t(a), n, dim=axis))
tile = lambda a, reps: Arr(_t(a).repeat(*reps))
broadcast_to = lambda a, s: Arr(torch.broadcast_to(_t(a), tuple(s)))
expand_dims = lambda a, axis: Arr(_t(a).unsqueeze(axis))
squeeze = lambda a, axis=None: Arr(_t(a).squeeze() if axis is None else _t(a).squeeze(axis))
reshape = lambda a, s: Arr(_t(a).reshape(*s))
transpose = lambda a, axes=None: a.transpose(*(axes or ()))
mean = lambda a, axis=None, keepdims=False: a.mean(axis, keepdims)
sum = lambda a, axis=None, keepdims=False: a.sum(axis, keepdims)  # noqa: A001
matmul = lambda a, b: Arr(_t(a) @ _t(b))
softmax = lambda a, axis=-1: Arr(torch.softmax(_t(a), dim=axis))
allclose = lambda a, b, rtol=1e-5, atol=1e-8: bool(torch.allclose(_t(a, b).float(), _t(b, a).float(), rtol=rtol, atol=atol))


def _mx_all(a, axis=None, keepdims=False):
    t = _t(a)
    return Arr(t.all() if axis is None else t.all(dim=axis, keepdim=keepdims))


def std(a, axis=None, keepdims=False):
    t = _t(a)
    return Arr(t.std(unbiased=False) if axis is None else t.std(dim=axis, keepdim=keepdims, unbiased=False))


def pad(a, widths, constant_values=0):
    t = _t(a)
    flat = []
    for lo, hi in reversed(list(widths)):
        flat += [lo, hi]
    return Arr(F.pad(t, flat, value=constant_values))


def conv2d(x, w, stride=1, padding=0, dilation=1, groups=1):
    """MLX layout: x (N,H,W,Cin), w (Cout,kH,kW,Cin) -> (N,H',W',Cout)."""
    y = F.conv2d(_t(x).permute(0, 3, 1, 2), _t(w).permute(0, 3, 1, 2), stride=stride, padding=padding, dilation=dilation, groups=groups)
    return Arr(y.permute(0, 2, 3, 1))


def compile(fn=None, **kw):  # noqa: A001
    return fn if fn is not None else (lambda f: f)


def eval(*a, **k):  # noqa: A001
    return None


class _Fast(types.ModuleType):
    @staticmethod
    def rms_norm(x, weight, eps):
        t = _t(x)
        y = t * torch.rsqrt(t.float().pow(2).mean(-1, keepdim=True) + eps).to(t.dtype)
        return Arr(y if weight is None else y * _t(weight))

    @staticmethod
    def scaled_dot_product_attention(q, k, v, scale=None, mask=None):
        q, k, v = _t(q), _t(k), _t(v)
        s = (q @ k.transpose(-1, -2)) * (scale if scale is not None else 1.0 / math.sqrt(q.shape[-1]))
        if mask is not None:
            s = s + _t(mask)
        return Arr(torch.softmax(s.float(), dim=-1).to(q.dtype) @ v)

    @staticmethod
    def metal_kernel(**kw):
        def _unavailable(*a, **k):
            raise RuntimeError("Metal kernels are not available in the shim")
        return _unavailable


class _Random(types.ModuleType):
    gen = torch.Generator().manual_seed(0)
    preset = []          # tensors to hand out (FIFO) instead of drawing: lets the pin script share noise with the oracle

    @classmethod
    def seed(cls, s):
        cls.gen.manual_seed(int(s))

    @classmethod
    def normal(cls, shape=(), dtype=float32, key=None, loc=0.0, scale=1.0):
        if cls.preset:
            t = cls.preset.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return Arr(t.clone())
        return Arr(torch.randn(tuple(shape), generator=cls.gen).to(_td(dtype)) * scale + loc)

    @classmethod
    def uniform(cls, low=0.0, high=1.0, shape=(), dtype=float32, key=None):
        return Arr((torch.rand(tuple(shape), generator=cls.gen) * (high - low) + low).to(_td(dtype)))

    @staticmethod
    def key(s):
        return Arr(torch.tensor([0, int(s)]))

    @staticmethod
    def split(k, num=2):
        return [Arr(k.t + i + 1) for i in range(num)]


# ------------------------------------------------------------------ mlx.nn
class Module:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise NotImplementedError

    def parameters(self):
        return {}

    def eval(self):
        return self

    def update(self, *a, **k):
        return self


class Linear(Module):
    def __init__(self, input_dims, output_dims, bias=True):
        self.weight = zeros((output_dims, input_dims))
        if bias:
            self.bias = zeros((output_dims,))

    def __call__(self, x):
        y = _t(x) @ _t(self.weight).to(_t(x).dtype).T
        if getattr(self, "bias", None) is not None:
            y = y + _t(self.bias).to(y.dtype)
        return Arr(y)


class LayerNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=True, bias=True):
        self.dims, self.eps, self.affine = dims, eps, affine
        if affine:
            self.weight, self.bias = ones((dims,)), zeros((dims,))

    def __call__(self, x):
        t = _t(x)
        y = F.layer_norm(t.float(), (self.dims,), eps=self.eps).to(t.dtype)
        if self.affine:
            y = y * _t(self.weight) + _t(self.bias)
        return Arr(y)


class RMSNorm(Module):
    def __init__(self, dims, eps=1e-5):
        self.weight, self.eps = ones((dims,)), eps

    def __call__(self, x):
        return _Fast.rms_norm(x, self.weight, self.eps)


class _Act(Module):
    fn = None

    def __call__(self, x):
        return type(self).fn(x)


silu = lambda x: Arr(F.silu(_t(x)))
gelu_approx = lambda x: Arr(F.gelu(_t(x), approximate="tanh"))
gelu = lambda x: Arr(F.gelu(_t(x)))
relu = lambda x: Arr(F.relu(_t(x)))


class SiLU(_Act):
    fn = staticmethod(silu)


class GELU(_Act):
    fn = staticmethod(gelu)


class ReLU(_Act):
    fn = staticmethod(relu)


class _Generic(Module):
    """Placeholder for layers outside the hot path that are only constructed, never called."""

    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k


def install():
    """Register fake `mlx`, `mlx.core`, `mlx.nn`, `mlx.utils` modules."""
    me = sys.modules[__name__]
    mlx = types.ModuleType("mlx")
    core = types.ModuleType("mlx.core")
    for k, v in vars(me).items():
        if not k.startswith("_") and k not in ("Module", "Linear", "LayerNorm", "RMSNorm", "SiLU", "GELU", "ReLU", "install"):
            setattr(core, k, v)
    core.fast = _Fast("mlx.core.fast")
    core.random = _Random("mlx.core.random")
    core.array = array
    core.all = _mx_all
    core.Dtype = Dtype
    core.metal = types.SimpleNamespace(clear_cache=lambda: None, is_available=lambda: False)
    core.clear_cache = lambda: None
    nn = types.ModuleType("mlx.nn")
    for k in ("Module", "Linear", "LayerNorm", "RMSNorm", "SiLU", "GELU", "ReLU", "silu", "gelu_approx", "gelu", "relu"):
        setattr(nn, k, getattr(me, k))
    for k in ("Conv1d", "Conv2d", "Conv3d", "ConvTranspose1d", "ConvTranspose2d", "GroupNorm", "Embedding", "Dropout", "Sequential", "Identity"):
        setattr(nn, k, _Generic)
    utils = types.ModuleType("mlx.utils")
    utils.tree_flatten = lambda t: []
    utils.tree_unflatten = lambda t: {}
    utils.tree_map = lambda f, t, *r: t
    mlx.core, mlx.nn, mlx.utils = core, nn, utils
    sys.modules.update({"mlx": mlx, "mlx.core": core, "mlx.nn": nn, "mlx.utils": utils})
    return core, nn
