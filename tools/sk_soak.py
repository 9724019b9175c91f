"""Soak test of the stream-K attention's in-launch hand-off: many launches while a second stream keeps other kernels in flight on
the same GPU (the situation of the AudioVideo engine's side stream), checked bit for bit against the first result, with the
sticky error word and the flags read back at the end.  usage: python tools/sk_soak.py [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N, D, H = 3456, 4096, 32
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(N, D, generator=g, device=dev).to(torch.bfloat16); k = torch.randn(N, D, generator=g, device=dev).to(torch.bfloat16)
vt = K.vt_transpose(torch.randn(N, D, generator=g, device=dev).to(torch.bfloat16), H)
ws = K.flash_attn_workspace(128, dev)
ref = K.flash_attn(q, k, vt, H, N, workspace=ws).clone()
side = torch.cuda.Stream()
a = torch.randn(3456, 4096, device=dev).to(torch.bfloat16); w = (torch.randn(4096, 4096, device=dev) / 64).to(torch.bfloat16)
x = torch.randn(3456, 4096, device=dev)
bad = 0
for i in range(n):
    with torch.cuda.stream(side):            # competing work: a full-chip GEMM, a bandwidth-bound norm, small kernels
        if i % 3 == 0: K.gemm(a, w)
        elif i % 3 == 1: K.adaln_rmsnorm(x)
        else: x.mul_(1.0)
    out = K.flash_attn(q, k, vt, H, N, workspace=ws)
    if i % 50 == 0:
        bad += int(not torch.equal(out, ref))
torch.cuda.synchronize()
flags = ws[:4096].view(torch.int32)
print(f"{n} stream-K launches beside a busy second stream: mismatches {bad}, flags nonzero {int((flags != 0).sum())}, error word {int(flags[1023])}")
