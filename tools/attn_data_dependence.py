"""Round 6: what the stale-maximum softmax of the attention kernel costs when the scores are NOT random-init shaped.
For the DiT's self-attention (3456 x 3456) and text cross-attention (3456 x 1024), bfloat16 and float16 builds: time per launch (best of 3 x 20) for
  random        q, k ~ N(0, 1): scores ~ N(0, 1) nats (what every other tool times)
  sink_last     one key of the LAST tile ~12 nats above every row's running maximum (an attention sink that the first tile does not contain)
  sink_first    the same key in the FIRST tile (the reference exponent then already covers it)
  ramp_x15      key norms growing x15 along the sequence (score spread 1 -> 15 nats)
  first_tile_0  the first tile's keys scaled to ~0 (reference exponent ~0, everything later sits up to ~5 nats above it)
and, with a counting build (LTX2HIP_LIB[_F16] = a library built with -DAT_COUNT_FALLBACK), the share of wave-tiles of the fast path that fell back
to the classic path.  Results are checked against an fp64 softmax on a sample of rows.
usage: python tools/attn_data_dependence.py [--f16]"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
dt = torch.float16 if "--f16" in sys.argv else torch.bfloat16
lib = nv.lib(dt)
counting = hasattr(lib, "ltx2_attn_fallback_counts")


def counts(reset=True):
    if not counting:
        return None
    a = (ctypes.c_ulonglong * 2)()
    torch.cuda.synchronize()
    lib.ltx2_attn_fallback_counts(a, 1 if reset else 0)
    return a[0], a[1]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def make(case, Nq, Nkv, H, hd, g):
    D = H * hd
    q = torch.randn(Nq, D, generator=g, device=dev)
    k = torch.randn(Nkv, D, generator=g, device=dev)
    v = torch.randn(Nkv, D, generator=g, device=dev)
    if case in ("sink_last", "sink_first"):
        # every query gets a common component along dim 0 of each head; the sink key is long along it: score = 4 * 44.4 / sqrt(128) ~ 15.7 nats, ~12 above the
        # maximum (~3.7 nats) of the other 3455 N(0, 1) scores
        row = Nkv - 5 if case == "sink_last" else 5
        q.view(Nq, H, hd)[:, :, 0] += 4.0
        k.view(Nkv, H, hd)[row] = 0.0
        k.view(Nkv, H, hd)[row, :, 0] = 44.4
    elif case == "ramp_x15":
        k *= (1.0 + 14.0 * torch.arange(Nkv, device=dev) / Nkv)[:, None]
    elif case == "first_tile_0":
        k[:64] *= 0.01
    return q.to(dt), k.to(dt), v.to(dt)


g = torch.Generator(device=dev).manual_seed(6)
print(f"dtype {dt}, counting build: {counting}", flush=True)
for (Nq, Nkv, H, hd) in [(3456, 3456, 32, 128), (3456, 1024, 32, 128)]:
    for case in ("random", "sink_last", "sink_first", "ramp_x15", "first_tile_0"):
        q, k, v = make(case, Nq, Nkv, H, hd, g)
        vt = K.vt_transpose(v, H, head_dim=hd)
        counts()
        out = K.flash_attn(q, k, vt, H, Nkv)
        c = counts()
        # fp64 check on 64 rows of 2 heads
        rows = torch.arange(0, Nq, Nq // 64, device=dev)[:64]
        err = 0.0
        for h in (0, H - 1):
            sl = slice(h * hd, (h + 1) * hd)
            s = (q[rows][:, sl].double() @ k[:, sl].double().T) / math.sqrt(hd)
            ref = torch.softmax(s, -1) @ v[:, sl].double()
            err = max(err, float((out[rows][:, sl].double() - ref).norm() / ref.norm()))
        best = min(timeit(lambda: K.flash_attn(q, k, vt, H, Nkv)) for _ in range(3))
        fb = "" if c is None else f"  fallback wave-tiles {c[1]}/{c[0]} = {100.0 * c[1] / max(c[0], 1):.2f} %"
        print(f"Nq={Nq} Nkv={Nkv} {case:13s}: {best * 1e6:7.1f} us  {4.0 * Nq * Nkv * H * hd / best / 1e12:6.1f} TF/s  rel-L2 vs fp64 {err:.2e}{fb}", flush=True)
