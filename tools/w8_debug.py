import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for (M, N, Kk, mode) in ((256, 256, 256, "rand"), (256, 256, 256, "ints"), (3456, 4096, 4096, "rand"), (70, 512, 256, "rand")):
    a = torch.randn(M, Kk, generator=g, device=dev).to(torch.bfloat16)
    if mode == "ints":
        w = torch.randint(-3, 4, (N, Kk), generator=g, device=dev).float()
        scale = torch.ones(N, device=dev)
    else:
        w = torch.randn(N, Kk, generator=g, device=dev) / Kk ** 0.5
        scale = torch.full((N,), float(w.abs().max() / 448.0), device=dev)
    codes = (w / scale[:, None]).to(torch.float8_e4m3fn)
    wdq = K.dequant_fp8(codes.view(torch.uint8), float(scale[0]))
    bias = torch.randn(N, generator=g, device=dev)
    for epi, b in ((nv.EPI_F32, bias), (nv.EPI_BF16, None), (nv.EPI_BF16, bias)):
        r2 = K.gemm(a, wdq, b, epilogue=epi)
        o2 = K.gemm_w8a16(a, codes.view(torch.uint8), scale, b, epilogue=epi)
        print("   epi", epi, "bias", b is not None, "mismatch", int((o2 != r2).sum()), "max diff", float((o2.float() - r2.float()).abs().max()))
    ref = K.gemm(a, wdq, None, epilogue=nv.EPI_F32)
    out = K.gemm_w8a16(a, codes.view(torch.uint8), scale, None, epilogue=nv.EPI_F32)
    d = (out - ref).abs()
    nbad = int((out != ref).sum())
    print(mode, M, N, Kk, "mismatch", nbad, "of", out.numel(), "max abs diff", float(d.max()), "ref max", float(ref.abs().max()))
    if nbad:
        idx = (out != ref).nonzero()[:8]
        print("   first bad (row, col):", idx.tolist())
        bad_cols = (out != ref).any(0).nonzero().flatten()
        bad_rows = (out != ref).any(1).nonzero().flatten()
        print("   bad cols", bad_cols.numel(), bad_cols[:16].tolist(), " bad rows", bad_rows.numel(), bad_rows[:16].tolist())
