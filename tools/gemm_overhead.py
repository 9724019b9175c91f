"""Fixed cost per output tile of the 4-wave GEMM: time M x N x K at several K (same M, N, epilogue) -> slope = time per K-tile round,
intercept = prologue + epilogue + launch per round of tiles.  bf16 and fp8."""
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
M = 3456
for name, Nn, epi in [("N=4096 bf16-out", 4096, nv.EPI_BF16), ("N=4096 resid", 4096, nv.EPI_RESID_GATE_F32), ("N=16384 gelu", 16384, nv.EPI_GELU_BF16), ("N=16384 bf16-out", 16384, nv.EPI_BF16),
                      ("N=12288 bf16-out", 12288, nv.EPI_BF16)]:
    rounds = math.ceil(math.ceil(M / 224) * (Nn // 256) / 256)
    res = {}
    for fp8 in (False, True):
        ts = []
        for Kk in (1024, 2048, 4096, 8192):
            a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
            w = (torch.randn(Nn, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
            b = torch.randn(Nn, device=dev)
            f32 = epi in (nv.EPI_F32, nv.EPI_RESID_GATE_F32)
            out = torch.zeros(M, Nn, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
            gt = torch.randn(Nn, device=dev) if f32 else None
            if fp8:
                a8, asc = K.quantize_rows_fp8(a); w8, wsc = K.quantize_rows_fp8(w)
                t = min(timeit(lambda: K.gemm_fp8(a8, asc, w8, wsc, b, epilogue=epi, out=out, gate_table=gt)) for _ in range(3))
            else:
                t = min(timeit(lambda: K.gemm(a, w, b, epilogue=epi, out=out, gate_table=gt)) for _ in range(3))
            ts.append((Kk, t))
        (k1, t1), (k2, t2) = ts[1], ts[3]
        slope = (t2 - t1) / (k2 - k1)
        res[fp8] = (ts, slope * 1024, t1 - slope * k1)
    for fp8 in (False, True):
        ts, per1k, icpt = res[fp8]
        print(f"{name:18s} {'fp8 ' if fp8 else 'bf16'} rounds={rounds}: " + " ".join(f"K={k}:{t:6.1f}us" for k, t in ts) + f" | {per1k:6.2f} us per 1024 K | intercept {icpt:6.1f} us = {icpt / rounds:5.1f} per round", flush=True)
