"""norm_mod (bf16 out) vs norm_mod with the fused per-token fp8 quantiser vs the stand-alone quantiser, DiT shapes."""
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
tab, emb = torch.randn(2, D, device=dev), torch.randn(2, D, device=dev)
kw = dict(scale_tab=tab[1], shift_tab=tab[0], scale_emb=emb[1], shift_emb=emb[0])
h = K.adaln_rmsnorm(x, **kw)
for name, k in (("modulated", kw), ("plain", {})):
    print(f"{name}: bf16 {timeit(lambda: K.adaln_rmsnorm(x, **k)):.1f} us | fused fp8 only {timeit(lambda: K.adaln_rmsnorm_fp8(x, want_bf16=False, **k)):.1f} us | "
          f"fused fp8 + bf16 {timeit(lambda: K.adaln_rmsnorm_fp8(x, **k)):.1f} us | quantise alone {timeit(lambda: K.quantize_rows_fp8(h)):.1f} us")
