"""Conv3d micro-timing on the VAE decoder's stage shapes (chunk of 7 latent frames at 768x512).
usage: python tools/conv_time.py   (env LTX2_GEMM_TILE=tall|notall|small|pp)"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
dev = torch.device("cuda:0")
shapes = [(49, 128, 192, 128, 128), (25, 64, 96, 256, 256), (13, 32, 48, 512, 512), (7, 16, 24, 1024, 1024), (49, 128, 192, 128, 48)]
for (T, H, W, Cin, Cout) in shapes:
    x = torch.randn(T, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 27, Cin, device=dev) / math.sqrt(27 * Cin)).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    for _ in range(2): K.conv3d(x, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): K.conv3d(x, w, b)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5 * 1e-3
    fl = 2.0 * 27 * Cin * Cout * T * H * W
    print(f"tile={os.environ.get('LTX2_GEMM_TILE','auto')} T={T} H={H} W={W} {Cin}->{Cout}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s")
