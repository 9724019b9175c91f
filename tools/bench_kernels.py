"""Micro-benchmarks of the hot kernels at the BASELINE shapes (768x512x65: N=3456, S=1024, D=4096).
Prints TFLOP/s (GEMM, attention, conv) per kernel, measured with HIP events on the launch stream.
Usage: python tools/bench_kernels.py [--quick]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K  # noqa: E402
from ltx_2_mlx_amd import _native as nv  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    quick = "--quick" in sys.argv
    N, S, D = 3456, 1024, 4096
    print(torch.cuda.get_device_name(0))
    shapes = [("qkv", N, 3 * D, D), ("o/q2", N, D, D), ("ff1", N, 4 * D, D), ("ff2", N, D, 4 * D), ("kv2", S, 2 * D, D), ("4096^3", 4096, 4096, 4096)]
    for name, M, Nn, Kk in shapes:
        a = torch.randn(M, Kk, device=dev).to(BF)
        w = (torch.randn(Nn, Kk, device=dev) / math.sqrt(Kk)).to(BF)
        b = torch.randn(Nn, device=dev)
        out = torch.empty(M, Nn, device=dev, dtype=BF)
        t = timeit(lambda: K.gemm(a, w, b, out=out))
        print(f"gemm {name:8s} M={M} N={Nn} K={Kk}: {t*1e6:9.1f} us  {2*M*Nn*Kk/t/1e12:7.1f} TF/s")
        ref = torch.empty(M, Nn, device=dev, dtype=BF)
        t2 = timeit(lambda: torch.matmul(a, w.t(), out=ref))
        print(f"   torch(hipBLASLt) reference:              {t2*1e6:9.1f} us  {2*M*Nn*Kk/t2/1e12:7.1f} TF/s")
    H = 32
    for name, nq, nkv in [("self", N, N), ("cross", N, S)]:
        q = torch.randn(nq, D, device=dev).to(BF)
        k = torch.randn(nkv, D, device=dev).to(BF)
        v = torch.randn(nkv, D, device=dev).to(BF)
        vt = K.vt_transpose(v, H)
        t = timeit(lambda: K.flash_attn(q, k, vt, H, nkv))
        print(f"attn {name:6s} Nq={nq} Nkv={nkv}: {t*1e6:9.1f} us  {4*nq*nkv*D/t/1e12:7.1f} TF/s")
        t = timeit(lambda: K.vt_transpose(v, H))
        print(f"   vt_transpose: {t*1e6:9.1f} us")
    x = torch.randn(N, D, device=dev)
    t = timeit(lambda: K.adaln_rmsnorm(x))
    print(f"rmsnorm N={N} D={D}: {t*1e6:8.1f} us  {N*D*6/t/1e9:7.1f} GB/s")
    tab = torch.randn(6, D, device=dev); emb = torch.randn(6, D, device=dev)
    t = timeit(lambda: K.adaln_rmsnorm(x, scale_tab=tab[1], shift_tab=tab[0], scale_emb=emb[1], shift_emb=emb[0]))
    print(f"adaln_rmsnorm N={N} D={D}: {t*1e6:8.1f} us  {N*D*6/t/1e9:7.1f} GB/s")
    qkv = torch.randn(N, 3 * D, device=dev).to(BF)
    wq = torch.ones(D, device=dev); cs = torch.randn(N, D // 2, device=dev)
    t = timeit(lambda: K.qknorm_rope_(qkv, D, 128, 0, wq, D, wq, 1e-6, cs, cs))
    print(f"qknorm_rope N={N} (q,k + rope): {t*1e6:8.1f} us  {(N*2*D*2*2 + N*D*4)/t/1e9:7.1f} GB/s")
    if quick:
        return
    for name, T, Hh, Ww, cin, cout in [("res1024", 7, 16, 24, 1024, 1024), ("res512", 13, 32, 48, 512, 512),
                                      ("res256", 25, 64, 96, 256, 256), ("res128", 49, 128, 192, 128, 128),
                                      ("out48", 49, 128, 192, 128, 48)]:
        xx = torch.randn(T, Hh, Ww, cin, device=dev).to(BF)
        w = (torch.randn(cout, 27, cin, device=dev) / math.sqrt(27 * cin)).to(BF)
        b = torch.randn(cout, device=dev)
        t = timeit(lambda: K.conv3d(xx, w, b), iters=5, warm=1)
        fl = 2.0 * 27 * cin * cout * T * Hh * Ww
        print(f"conv {name:8s} {T}x{Hh}x{Ww} {cin}->{cout}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    main()
