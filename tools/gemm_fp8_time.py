"""fp8 compute path at the DiT shapes: ltx2_gemm_fp8 (v_mfma_f32_32x32x64_f8f6f4) beside the bf16 kernel with the same epilogue, and the
per-token quantiser.  LTX2_F8_SCALED=1 times the v_mfma_scale_* form (unit block scales)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
N, D = 3456, 4096
print("scaled form" if os.environ.get("LTX2_F8_SCALED") == "1" else "plain form", torch.cuda.get_device_name(0))
for name, M, Nn, Kk, epi in [("qkv", N, 3 * D, D, nv.EPI_BF16), ("to_out", N, D, D, nv.EPI_RESID_GATE_F32), ("cross-q", N, D, D, nv.EPI_BF16),
                             ("ff1", N, 4 * D, D, nv.EPI_GELU_BF16), ("ff2", N, D, 4 * D, nv.EPI_RESID_GATE_F32), ("4096^3", 4096, 4096, 4096, nv.EPI_BF16)]:
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(Nn, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(Nn, device=dev)
    a8, asc = K.quantize_rows_fp8(a)
    w8, wsc = K.quantize_rows_fp8(w)
    f32 = epi in (nv.EPI_F32, nv.EPI_RESID_GATE_F32)
    out = torch.zeros(M, Nn, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    gt = torch.randn(Nn, device=dev) if epi == nv.EPI_RESID_GATE_F32 else None
    best = [1e9, 1e9, 1e9]
    for _ in range(3):
        best[0] = min(best[0], timeit(lambda: K.gemm(a, w, b, epilogue=epi, out=out, gate_table=gt)))
        best[1] = min(best[1], timeit(lambda: K.gemm_fp8(a8, asc, w8, wsc, b, epilogue=epi, out=out, gate_table=gt)))
        best[2] = min(best[2], timeit(lambda: K.quantize_rows_fp8(a)))
    fl = 2.0 * M * Nn * Kk
    print(f"{name:8s} M={M} N={Nn} K={Kk}: bf16 {best[0]*1e6:7.1f} us {fl/best[0]/1e12:7.1f} TF/s | fp8 {best[1]*1e6:7.1f} us {fl/best[1]/1e12:7.1f} TF/s "
          f"({best[0]/best[1]:.2f}x) | quantise A {best[2]*1e6:6.1f} us ({M*Kk*3/best[2]/1e9:.0f} GB/s)", flush=True)
