"""Is the dominant GEMM's fabric over-fetch (FETCH_SIZE 1.8-2.0x the algorithmic bytes: every XCD's L2 fetches the panels its tiles need) served by the
Infinity Cache or by HBM, and does it matter?  The TCC counters sit on the L2 side of the fabric and cannot tell (TCC_EA0_RDREQ_DRAM counts requests
routed to the local memory controller, MALL hit or not), so this measures the consequence instead: the same launch on operands that were just
touched (the working set of one launch, 175 / 360 MB, cycled alone: as MALL-warm as it gets) against a rotation over enough distinct operand sets
(>= 1.5 GB) that every launch finds the Infinity Cache holding only other launches' data (first-touch panels come from HBM; re-fetches by the other
XCDs inside the launch can still hit what the first XCD brought in)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
def timeit(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, Kk) in [(3456, 4096, 4096), (3456, 4096, 16384)]:
    per_set = (M * Kk + N * Kk) * 2 + 2 * M * N * 4
    nset = max(2, int(math.ceil(1.6e9 / per_set)))
    sets = []
    for i in range(nset):
        sets.append((torch.randn(M, Kk, device=dev).to(torch.bfloat16), (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16),
                     torch.zeros(M, N, device=dev)))
    b = torch.randn(N, device=dev); gate = 0.01 * torch.randn(N, device=dev)
    def run(i, rot):
        a, w, x = sets[i % nset if rot else 0]
        K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate_table=gate)
    best = [1e9, 1e9]
    for _ in range(3):
        best[0] = min(best[0], timeit(lambda i: run(i, False), 4 * nset))
        best[1] = min(best[1], timeit(lambda i: run(i, True), 4 * nset))
    print(f"M={M} N={N} K={Kk}: one operand set ({per_set/1e6:.0f} MB) re-used {best[0]:7.1f} us | rotating over {nset} sets ({nset*per_set/1e9:.2f} GB) {best[1]:7.1f} us "
          f"| {best[1]/best[0]-1:+.1%}", flush=True)
