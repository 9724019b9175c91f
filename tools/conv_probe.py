"""One conv3d shape a few times (for rocprofv3 --pmc).  usage: conv_probe.py T H W Cin Cout [iters]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
T, H, W, Cin, Cout = (int(x) for x in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
x = torch.randn(T, H, W, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(Cout, 27, Cin, device=dev) / math.sqrt(27 * Cin)).to(torch.bfloat16)
b = torch.randn(Cout, device=dev)
for _ in range(iters):
    K.conv3d(x, w, b)
torch.cuda.synchronize()
