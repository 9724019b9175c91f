"""Repeat each hot kernel on fixed inputs and compare outputs bit-for-bit (none of them uses atomics)."""
import math, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rep(name, fn, n=30):
    ref = fn().clone()
    bad = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(n))
    print(f"{name}: {bad}/{n} runs differ")
for (Nq, Nkv, H, hd) in [(3456, 3456, 32, 128), (3456, 1024, 32, 128), (3456, 68, 32, 64), (300, 1000, 4, 128)]:
    D = H * hd
    q = torch.randn(Nq, D, device=dev).to(torch.bfloat16); k = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(Nkv, D, device=dev).to(torch.bfloat16); vt = K.vt_transpose(v, H, head_dim=hd)
    rep(f"attn {Nq}x{Nkv} H{H} hd{hd}", lambda: K.flash_attn(q, k, vt, H, Nkv))
for (M, N, Kk) in [(1024, 8192, 3840), (1024, 4096, 4096), (3456, 4096, 4096), (3456, 16384, 4096), (300, 512, 256)]:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16); w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rep(f"gemm {M}x{N}x{Kk}", lambda: K.gemm(a, w, b, out=out))
x = torch.randn(13, 32, 48, 256, device=dev).to(torch.bfloat16); w = (torch.randn(256, 27, 256, device=dev) / 83).to(torch.bfloat16); b = torch.randn(256, device=dev)
rep("conv3d 13x32x48 256->256", lambda: K.conv3d(x, w, b))
x = torch.randn(9, 64, 96, 128, device=dev).to(torch.bfloat16); w = (torch.randn(128, 27, 128, device=dev) / 58).to(torch.bfloat16); b = torch.randn(128, device=dev)
rep("conv3d 9x64x96 128->128", lambda: K.conv3d(x, w, b))
