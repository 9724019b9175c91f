"""Round 6: what the folded pre-norm's halves cost the GEMMs they ride on (same box, best of 3 x 20 launches, us): the consumer shapes (plain kernel | VAR-30 kernel
with row factors), the producer shapes (plain gated-residual epilogue | + shadow and partial sums), and producer -> norm -> consumer against producer -> consumer
on dependent data."""
import os, sys, torch
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
best = lambda fn: min(timeit(fn) for _ in range(3))
M, D = 3456, 4096
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
a = torch.randn(M, D, generator=g, device=dev).to(bf)
ss = (torch.rand(D // 256, 3712, generator=g, device=dev) * 256).float()
for name, NO, epi in (("cross-Q", D, nv.EPI_BF16), ("QKV", 3 * D, nv.EPI_BF16), ("FFN-up", 4 * D, nv.EPI_GELU_BF16)):
    w = (0.02 * torch.randn(NO, D, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(NO, generator=g, device=dev)
    out = torch.empty(M, NO, device=dev, dtype=bf)
    t0 = best(lambda: K.gemm(a, w, b, epi, out=out))
    t_rf = best(lambda: K.gemm_fold(a, w, b, epi, out=out, rf_parts=ss, rf_dim=D))
    print(f"{name:8s} plain {t0:7.1f} | row factors {t_rf:7.1f} ({t_rf - t0:+.1f})", flush=True)
for name, Kd in (("to_out", D), ("FFN-down", 4 * D)):
    att = torch.randn(M, Kd, generator=g, device=dev).to(bf)
    w = (0.02 * torch.randn(D, Kd, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(D, generator=g, device=dev)
    gate = torch.ones(D, device=dev)
    x = torch.randn(M, D, generator=g, device=dev)
    y = torch.zeros(M, D, device=dev, dtype=bf)
    sc = 0.1 * torch.randn(D, generator=g, device=dev)
    t0 = best(lambda: K.gemm(att, w, b, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate))
    t1 = best(lambda: K.gemm_fold(att, w, b, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate, shadow=y, shadow_scale=sc))
    print(f"{name:8s} plain {t0:7.1f} | shadow + sums {t1:7.1f} ({t1 - t0:+.1f})", flush=True)
print("pairs (us per pair):", flush=True)
for name, Kp, NO, epi in (("to_out -> cross-Q", D, D, nv.EPI_BF16),):
    att = torch.randn(M, Kp, generator=g, device=dev).to(bf)
    wo = (0.02 * torch.randn(D, Kp, generator=g, device=dev)).to(bf)
    bo = 0.02 * torch.randn(D, generator=g, device=dev)
    gate = torch.full((D,), 1e-3, device=dev)          # (small: x stays bounded over the timing loop)
    x = torch.randn(M, D, generator=g, device=dev)
    y = torch.zeros(M, D, device=dev, dtype=bf)
    w = (0.02 * torch.randn(NO, D, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(NO, generator=g, device=dev)
    out = torch.empty(M, NO, device=dev, dtype=bf)
    def classic():
        K.gemm(att, wo, bo, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate)
        h = K.adaln_rmsnorm(x, 1e-6)
        K.gemm(h, w, b, epi, out=out)
    def folded():
        r = K.gemm_fold(att, wo, bo, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate, shadow=y)
        K.gemm_fold(y, w, b, epi, out=out, rf_parts=r[1], rf_dim=D)
    tc, tf = best(classic), best(folded)
    print(f"{name:20s} classic {tc:7.1f} | folded {tf:7.1f} ({tf - tc:+.1f})", flush=True)
