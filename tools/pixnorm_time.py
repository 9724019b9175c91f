"""pixnorm_mod_silu bandwidth on the VAE decoder's stage shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
for (P, C) in [(49 * 128 * 192, 128), (25 * 64 * 96, 256), (13 * 32 * 48, 512), (7 * 16 * 24, 1024)]:
    x = torch.randn(P, C, device=dev).to(torch.bfloat16)
    tab = 0.2 * torch.randn(4, C, device=dev); te = 0.2 * torch.randn(4 * C, device=dev)
    for _ in range(3): K.pixnorm_mod_silu(x, tab, te, 0, 1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): K.pixnorm_mod_silu(x, tab, te, 0, 1)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"P={P} C={C}: {t*1e6:7.1f} us  {4.0 * P * C / t / 1e12:5.2f} TB/s")
