import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(512, dtype=torch.int64, device=dev)
os.environ["LTX2_PP_DBG"] = hex(dbg.data_ptr())
import ltx_2_mlx_amd.kernels as K
M = N = 4096; Kk = 4096
a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3): K.gemm(a, w, None, out=out)
torch.cuda.synchronize()
d = dbg.cpu().reshape(2, 256)
base = int(min(d[0, 0], d[1, 0]))
names = ["La_start", "La_rd_issued", "La_lgkm0", "Ma_start", "Ma_end", "Lb_start", "Lb_vmcnt2", "Lb_lgkm0", "Mb_start", "Mb_mfma_end", "Mb_vmcnt6"]
for g in range(2):
    print(f"group {g} (wave {g*4}):")
    prev = None
    for i in range(33):
        v = int(d[g, i]) - base
        print(f"   t={8 + i // 11} {names[i % 11]:12s} {v:7d}" + (f"  (+{v - prev})" if prev is not None else ""))
        prev = v
