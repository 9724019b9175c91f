"""norm_mod (AdaLN RMSNorm) and qknorm_rope at the DiT's shapes: N = 3456 rows, D = 4096."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
N, D = 3456, 4096
x = torch.randn(N, D, device=dev)
st, sh, se, he = (0.1 * torch.randn(D, device=dev) for _ in range(4))
print(f"norm_mod plain      {timeit(lambda: K.adaln_rmsnorm(x)):6.1f} us")
print(f"norm_mod modulated  {timeit(lambda: K.adaln_rmsnorm(x, 1e-6, False, st, sh, se, he, 0)):6.1f} us   (table + embedding rows in two parts: four vectors per block)")
print(f"norm_mod modulated  {timeit(lambda: K.adaln_rmsnorm(x, 1e-6, False, st, sh)):6.1f} us   (combined rows: two vectors per block -- the engine's form since round 4)")
emb = 0.1 * torch.randn(N, 2 * D, device=dev)
print(f"norm_mod per-token  {timeit(lambda: K.adaln_rmsnorm(x, 1e-6, False, st, sh, emb[:, :D], emb[:, D:], 2 * D)):6.1f} us")
qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16)
qw, kw = torch.ones(D, device=dev), torch.ones(D, device=dev)
cos, sin = torch.rand(N, D // 2, device=dev), torch.rand(N, D // 2, device=dev)
print(f"qknorm_rope q+k     {timeit(lambda: K.qknorm_rope_(qkv, D, 128, 0, qw, D, kw, 1e-6, cos, sin)):6.1f} us")
print(f"qknorm q only       {timeit(lambda: K.qknorm_rope_(qkv, D, 128, 0, qw)):6.1f} us")
