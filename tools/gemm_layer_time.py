"""The six GEMMs of one DiT layer with their real epilogues beside the plain bf16 epilogue (same box): what each fused epilogue costs."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = int(os.environ.get("M", 3456))
tot = {}
for name, N, Kk, epi in [("qkv", 12288, 4096, nv.EPI_BF16), ("attn_out", 4096, 4096, nv.EPI_RESID_GATE_F32), ("cross_q", 4096, 4096, nv.EPI_BF16),
                         ("cross_out", 4096, 4096, nv.EPI_RESID_GATE_F32), ("ffn_up", 16384, 4096, nv.EPI_GELU_BF16), ("ffn_down", 4096, 16384, nv.EPI_RESID_GATE_F32)]:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.zeros(M, N, device=dev)
    gate = torch.randn(N, device=dev)
    plain = lambda: K.gemm(a, w, b, out=ob)
    if epi == nv.EPI_RESID_GATE_F32:
        real = lambda: K.gemm(a, w, b, epilogue=epi, out=x, gate_table=gate)
    else:
        real = lambda: K.gemm(a, w, b, epilogue=epi, out=ob)
    t0 = min(timeit(plain) for _ in range(3)); t1 = min(timeit(real) for _ in range(3))
    fl = 2.0 * M * N * Kk
    tot[name] = t1
    print(f"{name:10s} M={M} N={N} K={Kk}: bf16 epilogue {t0:7.1f} us {fl/t0/1e9:7.1f} TF/s | real epilogue {t1:7.1f} us {fl/t1/1e9:7.1f} TF/s | +{t1-t0:5.1f} us", flush=True)
print("layer GEMM total", sum(tot.values()), "us")
