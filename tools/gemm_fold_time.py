"""Round 6: what the folded-norm halves cost the GEMMs they ride on (same box, best of 3 x 20 launches, us): the DiT's three consumer shapes (plain kernel |
VAR-30 kernel + row factors | + extra row | both) and its producer shapes (plain gated-residual epilogue | + shadow and partial sums | + extra row copy)."""
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
a1 = torch.randn(M + 1, D, generator=g, device=dev).to(bf)
a = a1[:M]
ss = (torch.rand(D // 256, 3712, generator=g, device=dev) * 256).float()
for name, NO, epi in (("cross-Q", D, nv.EPI_BF16), ("QKV", 3 * D, nv.EPI_BF16), ("FFN-up", 4 * D, nv.EPI_GELU_BF16)):
    w = (0.02 * torch.randn(NO, D, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(NO, generator=g, device=dev)
    out = torch.empty(M, NO, device=dev, dtype=bf)
    t0 = best(lambda: K.gemm(a, w, b, epi, out=out))
    t_rf = best(lambda: K.gemm_fold(a, w, b, epi, out=out, rf_parts=ss, rf_dim=D))
    t_x = best(lambda: K.gemm_fold(a1, w, b, epi, out=out, xrow=True, xrow_bias=b))
    t_b = best(lambda: K.gemm_fold(a1, w, b, epi, out=out, rf_parts=ss, rf_dim=D, xrow=True, xrow_bias=b))
    print(f"{name:8s} plain {t0:7.1f} | row factors {t_rf:7.1f} ({t_rf - t0:+.1f}) | extra row {t_x:7.1f} ({t_x - t0:+.1f}) | both {t_b:7.1f} ({t_b - t0:+.1f})", flush=True)
for name, Kd in (("to_out", D), ("FFN-down", 4 * D)):
    att = torch.randn(M, Kd, generator=g, device=dev).to(bf)
    w = (0.02 * torch.randn(D, Kd, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(D, generator=g, device=dev)
    gate = torch.ones(D, device=dev)
    x = torch.randn(M, D, generator=g, device=dev)
    y = torch.zeros(M + 1, D, device=dev, dtype=bf)
    sc = 0.1 * torch.randn(D, generator=g, device=dev)
    tn = torch.randn(D, generator=g, device=dev).to(bf)
    t0 = best(lambda: K.gemm(att, w, b, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate))
    t1 = best(lambda: K.gemm_fold(att, w, b, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate, shadow=y, shadow_scale=sc))
    t2 = best(lambda: K.gemm_fold(att, w, b, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate, shadow=y, shadow_scale=sc, shadow_xrow=tn))
    print(f"{name:8s} plain {t0:7.1f} | shadow + sums {t1:7.1f} ({t1 - t0:+.1f}) | + extra row {t2:7.1f} ({t2 - t0:+.1f})", flush=True)

# ---- producer -> consumer PAIRS on dependent data (what the step runs): classic = gated-residual GEMM, norm pass, projection; folded = the two GEMMs alone
print("pairs (us per pair):", flush=True)
for name, Kp, NO, epi in (("to_out -> cross-Q", D, D, nv.EPI_BF16), ("to_out2 -> FFN-up", D, 4 * D, nv.EPI_GELU_BF16), ("FFN-down -> QKV", 4 * D, 3 * D, nv.EPI_BF16)):
    att = torch.randn(M, Kp, generator=g, device=dev).to(bf)
    wo = (0.02 * torch.randn(D, Kp, generator=g, device=dev)).to(bf)
    bo = 0.02 * torch.randn(D, generator=g, device=dev)
    gate = torch.full((D,), 1e-3, device=dev)          # (small: x stays bounded over the timing loop)
    x = torch.randn(M, D, generator=g, device=dev)
    y = torch.zeros(M + 1, D, device=dev, dtype=bf)
    sc = 0.1 * torch.randn(D, generator=g, device=dev)
    sh = 0.1 * torch.randn(D, generator=g, device=dev)
    tn = torch.randn(D, generator=g, device=dev).to(bf)
    w = (0.02 * torch.randn(NO, D, generator=g, device=dev)).to(bf)
    b = 0.02 * torch.randn(NO, generator=g, device=dev)
    out = torch.empty(M, NO, device=dev, dtype=bf)
    ssb = torch.zeros(D // 256, 3712, device=dev)
    def classic():
        K.gemm(att, wo, bo, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate)
        h = K.adaln_rmsnorm(x, 1e-6, scale_tab=sc, shift_tab=sh)
        K.gemm(h, w, b, epi, out=out)
    def folded():
        r = K.gemm_fold(att, wo, bo, nv.EPI_RESID_GATE_F32, out=x, gate_table=gate, shadow=y, shadow_scale=sc, shadow_xrow=tn)
        K.gemm_fold(y, w, b, epi, out=out, rf_parts=r[1], rf_dim=D, xrow=True, xrow_bias=b)
    tc, tf = best(classic), best(folded)
    print(f"{name:20s} classic {tc:7.1f} | folded {tf:7.1f} ({tf - tc:+.1f})", flush=True)
