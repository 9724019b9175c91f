"""One attention shape a few times (for rocprofv3 --pmc).  usage: attn_probe.py Nq Nkv H hd [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
Nq, Nkv, H, hd = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
q = torch.randn(Nq, H * hd, device=dev).to(torch.bfloat16)
k = torch.randn(Nkv, H * hd, device=dev).to(torch.bfloat16)
v = torch.randn(Nkv, H * hd, device=dev).to(torch.bfloat16)
vt = K.vt_transpose(v, H, head_dim=hd)
for _ in range(iters):
    K.flash_attn(q, k, vt, H, Nkv)
torch.cuda.synchronize()
