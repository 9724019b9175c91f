"""Does reading the NEXT GEMM's weights into the Infinity Cache from a side stream, while the current kernel runs, shorten that GEMM?  (round 5; tools/gemm_mall_probe.py
measured +14 % for a K = 4096 gated-residual GEMM on all-cold operands.)  Rotation over >= 1.6 GB of weight sets, as the step sees them: activations and the residual just
written (warm), weights cold.  Per iteration: [touch(W of set i + 1) on the side stream with `blocks` workgroups] beside [GEMM(set i)] on the main stream.
usage: python tools/gemm_prefetch_probe.py   (needs ltx-2-mlx_amd/lib/ab/touch.so: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/touch.hip -o ...)"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
T = ctypes.CDLL(os.path.join(ROOT, "ltx-2-mlx_amd", "lib", "ab", "touch.so"))
T.touch_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def timeit(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, Kk, epi) in [(3456, 4096, 4096, nv.EPI_RESID_GATE_F32), (3456, 4096, 4096, nv.EPI_BF16), (3456, 12288, 4096, nv.EPI_BF16), (3456, 16384, 4096, nv.EPI_GELU_BF16)]:
    wbytes = N * Kk * 2
    nset = max(3, int(math.ceil(1.6e9 / wbytes)))
    ws = [(torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16) for _ in range(nset)]
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    x = torch.zeros(M, N, device=dev) if epi == nv.EPI_RESID_GATE_F32 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev); gate = 0.01 * torch.randn(N, device=dev)
    def gemm(w):
        if epi == nv.EPI_RESID_GATE_F32: K.gemm(a, w, b, epilogue=epi, out=x, gate_table=gate)
        else: K.gemm(a, w, b, epilogue=epi, out=x)
    def run(i, mode, blocks):
        w = ws[i % nset] if mode else ws[0]
        if mode == 2:
            nxt = ws[(i + 1) % nset]
            side.wait_stream(torch.cuda.current_stream())            # the touch starts when the previous GEMM is done, i.e. beside THIS one
            with torch.cuda.stream(side):
                T.touch_launch(nxt.data_ptr(), wbytes, blocks, sink.data_ptr(), side.cuda_stream)
        gemm(w)
        if mode == 2:
            torch.cuda.current_stream().wait_stream(side)
    res = {}
    for _ in range(3):
        for tag, mode, blocks in [("warm", 0, 0), ("cold W", 1, 0), ("touch 16", 2, 16), ("touch 32", 2, 32), ("touch 64", 2, 64)]:
            t = timeit(lambda i: run(i, mode, blocks), 3 * nset)
            res[tag] = min(res.get(tag, 1e9), t)
    print(f"M={M} N={N} K={Kk} epi={epi}: " + " | ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
