import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(512, dtype=torch.int64, device=dev)
os.environ["LTX2_ATTN_DBG"] = hex(dbg.data_ptr())
import ltx_2_mlx_amd.kernels as K
H, N, D = 32, 3456, 4096
q = torch.randn(N, D, device=dev).to(torch.bfloat16); k = torch.randn(N, D, device=dev).to(torch.bfloat16); v = torch.randn(N, D, device=dev).to(torch.bfloat16)
vt = K.vt_transpose(v, H)
for _ in range(3): K.flash_attn(q, k, vt, H, N)
torch.cuda.synchronize()
d = dbg.cpu().reshape(2, 256)
base = int(min(d[0, 0], d[1, 0]))
names = ["V_start", "V_end", "M_start", "M_mfma_done", "M_end(vmcnt)"]
for g in range(2):
    print(f"group {g}:")
    prev = None
    for i in range(15):
        v_ = int(d[g, i]) - base
        print(f"   t={8 + i // 5} {names[i % 5]:12s} {v_:7d}" + (f"  (+{v_ - prev})" if prev is not None else ""))
        prev = v_
