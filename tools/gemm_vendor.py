"""Yardstick only (never on the product path): hipBLASLt/rocBLAS through torch.matmul on the DiT's GEMM shapes,
beside our kernels.  usage: python tools/gemm_vendor.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
shapes = [(3456, 4096, 4096), (3456, 12288, 4096), (3456, 16384, 4096), (3456, 4096, 16384), (4096, 4096, 4096), (8192, 8192, 8192)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for (M, N, Kk) in shapes:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_ours = t_vendor = 1e9          # alternate the two and keep the best of three: whichever runs first after an allocation reads 5-10 % slow
    for _ in range(3):
        t_ours = min(t_ours, timeit(lambda: K.gemm(a, w, b, out=out)))
        t_vendor = min(t_vendor, timeit(lambda: torch.matmul(a, w.t(), out=out)))
    fl = 2 * M * N * Kk
    print(f"M={M} N={N} K={Kk}: ours {t_ours*1e6:8.1f} us {fl/t_ours/1e12:7.1f} TF/s | torch.matmul {t_vendor*1e6:8.1f} us {fl/t_vendor/1e12:7.1f} TF/s")
