"""Run one GEMM shape a few times (for rocprofv3 --pmc). Usage: gemm_probe.py M N K [iters]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
M, N, Kk = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
gate = torch.randn(1, N, device=dev)
for _ in range(iters):
    K.gemm(a, w, b, out=out, epilogue=epi, gate=gate if epi == 4 else None, gate_table=gate[0] if epi == 4 else None)
torch.cuda.synchronize()
