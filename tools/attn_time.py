"""The attention kernel on the DiT shapes (same box, best of 3): us and TF/s.  Same-box A/B of two builds: tools/ab_run.sh 2 "base NAME" python tools/attn_time.py"""
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
SHAPES = [(3456, 3456, 32, 128), (3456, 1024, 32, 128), (13824, 13824, 32, 128), (13824, 1024, 32, 128), (3456, 68, 32, 64), (68, 3456, 32, 64)]
for (Nq, Nkv, H, hd) in SHAPES[:int(sys.argv[1]) if len(sys.argv) > 1 else None]:        # optional argument: only the first n shapes
    D = H * hd
    q = torch.randn(Nq, D, device=dev).to(torch.bfloat16)
    k = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(Nkv, D, device=dev).to(torch.bfloat16)
    vt = K.vt_transpose(v, H, head_dim=hd)
    best = min(timeit(lambda: K.flash_attn(q, k, vt, H, Nkv)) for _ in range(3))
    fl = 4.0 * Nq * Nkv * D
    print(f"Nq={Nq} Nkv={Nkv} H={H} hd={hd}: {best*1e6:8.1f} us {fl/best/1e12:7.1f} TF/s", flush=True)
