"""Yardstick only (never on the product path): torch SDPA (CK / AOTriton flash attention) beside attn_fwd_kernel."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for (Nq, Nkv, H, hd) in [(3456, 3456, 32, 128), (3456, 1024, 32, 128), (13824, 13824, 32, 128), (3456, 68, 32, 64), (68, 3456, 32, 64)]:
    D = H * hd
    q = torch.randn(Nq, D, device=dev).to(torch.bfloat16)
    k = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    vt = K.vt_transpose(v, H, head_dim=hd)
    t_ours = timeit(lambda: K.flash_attn(q, k, vt, H, Nkv))
    q4, k4, v4 = (x.reshape(1, -1, H, hd).transpose(1, 2) for x in (q, k, v))
    try:
        t_v = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4))
    except Exception as ex:
        t_v = float("nan")
    fl = 4.0 * Nq * Nkv * D
    print(f"Nq={Nq} Nkv={Nkv} H={H} hd={hd}: ours {t_ours*1e6:8.1f} us {fl/t_ours/1e12:7.1f} TF/s | torch SDPA {t_v*1e6:8.1f} us {fl/t_v/1e12:7.1f} TF/s")
