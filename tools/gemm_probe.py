"""Run one GEMM shape a few times (for rocprofv3 --pmc). Usage: gemm_probe.py M N K [iters]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
M, N, Kk = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    K.gemm(a, w, b, out=out)
torch.cuda.synchronize()
