"""Stream-K attention beside the plain grid on the DiT shapes (same box, alternating), with the max |difference|."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for (Nq, Nkv, H, hd) in [(3456, 3456, 32, 128), (3456, 1024, 32, 128), (13824, 13824, 32, 128), (13824, 1024, 32, 128), (3456, 68, 32, 64), (68, 3456, 32, 64)]:
    D = H * hd
    q = torch.randn(Nq, D, device=dev).to(torch.bfloat16)
    k = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    vt = K.vt_transpose(v, H, head_dim=hd)
    ws = K.flash_attn_workspace(hd, dev)
    a = K.flash_attn(q, k, vt, H, Nkv); b = K.flash_attn(q, k, vt, H, Nkv, workspace=ws)
    diff = (a.float() - b.float()).abs().max().item()
    best = [1e9, 1e9]
    for _ in range(3):
        best[0] = min(best[0], timeit(lambda: K.flash_attn(q, k, vt, H, Nkv)))
        best[1] = min(best[1], timeit(lambda: K.flash_attn(q, k, vt, H, Nkv, workspace=ws)))
    fl = 4.0 * Nq * Nkv * D
    print(f"Nq={Nq} Nkv={Nkv} H={H} hd={hd}: plain {best[0]*1e6:8.1f} us {fl/best[0]/1e12:7.1f} TF/s | stream-K {best[1]*1e6:8.1f} us {fl/best[1]/1e12:7.1f} TF/s | max diff {diff:.3g}", flush=True)
