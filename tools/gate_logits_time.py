import os, sys, torch
sys.path.insert(0, "/root/repo")
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
H, hd, Nq, Nkv, Dq = 32, 128, 3456, 64, 4096
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
q = torch.randn(Nq, H * hd, generator=g, device=dev).to(bf); k = torch.randn(Nkv, H * hd, generator=g, device=dev).to(bf); v = torch.randn(Nkv, H * hd, generator=g, device=dev).to(bf)
x = torch.randn(Nq, Dq, generator=g, device=dev).to(bf); wg = (0.02 * torch.randn(H, Dq, generator=g, device=dev)).to(bf); bg = torch.zeros(H, device=dev)
vt = K.vt_transpose(v, H)
lg = torch.zeros(Nq, H, device=dev)
t_both = min(timeit(lambda: K.flash_attn_gated_parts(q, k, vt, H, Nkv, x, wg, bg)) for _ in range(3))
t_att = min(timeit(lambda: K.flash_attn_gated(q, k, vt, H, Nkv, lg)) for _ in range(3))
print(f"gate logits (8 K-slice parts) 3456 x 4096 -> 32: {t_both - t_att:.1f} us (gated attention with / without the logit kernel: {t_both:.1f} / {t_att:.1f})")
