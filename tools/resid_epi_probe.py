"""Why does the gated fp32-residual epilogue cost more behind K = 16384 than behind K = 4096?  Time the GEMM alone (events around each
launch) with the residual tile (a) cold: last touched before the operands streamed through the caches, (b) re-touched right before."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
M = 3456
for N, Kk in [(4096, 4096), (4096, 16384)]:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1
    x = torch.zeros(M, N, device=dev)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    gate = torch.randn(N, device=dev)
    def run(touch, epi_resid, n=30):
        ts = []
        for i in range(n + 5):
            if touch:
                x.add_(0.0)          # read + write x: in L2 / MALL again
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            if epi_resid:
                K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate_table=gate)
            else:
                K.gemm(a, w, b, out=ob)
            e.record()
            torch.cuda.synchronize()
            if i >= 5: ts.append(s.elapsed_time(e) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    print(f"N={N} K={Kk}: bf16 {run(False, False):7.1f} us | resid cold {run(False, True):7.1f} us | resid after touching x {run(True, True):7.1f} us | bf16 after touching x {run(True, False):7.1f}", flush=True)
