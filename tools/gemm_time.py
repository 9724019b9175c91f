"""GEMM micro-timing on the DiT shapes.  usage: python tools/gemm_time.py [M N K]   (env LTX2_GEMM_TILE, LTX2_PP_BM)"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
shapes = [(3456, 4096, 4096), (3456, 12288, 4096), (3456, 16384, 4096), (3456, 4096, 16384), (4096, 4096, 4096), (8192, 8192, 8192)]
if len(sys.argv) > 3:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
for (M, N, Kk) in shapes:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): K.gemm(a, w, b, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): K.gemm(a, w, b, out=out)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"bm={os.environ.get('LTX2_PP_BM','auto')} tile={os.environ.get('LTX2_GEMM_TILE','auto')} M={M} N={N} K={Kk}: {t*1e6:8.1f} us {2*M*N*Kk/t/1e12:7.1f} TF/s")
