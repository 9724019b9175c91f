"""Epilogue cost on the DiT's gated-residual GEMM shapes: EPI_BF16 vs EPI_RESID_GATE_F32 (x += gate * (acc + bias), fp32 x)."""
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
for (M, N, Kk) in [(3456, 4096, 4096), (3456, 4096, 16384)]:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.zeros(M, N, device=dev)
    gate = 0.01 * torch.randn(1, N, device=dev)      # small: x += gate * (.) over the timing loop stays finite (the socket is power-limited: saturated operands run faster)
    t0 = timeit(lambda: K.gemm(a, w, b, out=ob))
    x.zero_()
    t1 = timeit(lambda: K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate_table=gate[0]))
    t2 = timeit(lambda: K.gemm(a, w, b, epilogue=nv.EPI_F32, out=x))
    x.zero_()
    t3 = timeit(lambda: K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate=gate, gate_table=gate[0]))
    x.zero_()
    t1b = timeit(lambda: K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate_table=gate[0]))      # table + a row-invariant gate row read per row (the engine's shared-AdaLN call before round 4)
    print(f"M={M} N={N} K={Kk}: bf16 out {t0:7.1f} us | f32 out {t2:7.1f} us | gated fp32 residual, table only {t1:7.1f} us | table + stride-0 gate row {t3:7.1f} us | table only again {t1b:7.1f} us")
