"""One multi-round launch vs power-of-two column chunks (each launch Nt = 16/32/64.. column tiles).
Result on MI355X: no difference once the two forms are alternated and the best of three is taken (a first pass that timed
the single launch first showed it 8-14 % slower at Nt = 48/80: clock / warm-up order, not the kernel)."""
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
    return s.elapsed_time(e) / n * 1e3
M, Kk = 3456, 4096
if len(sys.argv) > 2: M, Kk = int(sys.argv[1]), int(sys.argv[2])
for N in (4096, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576):
    a = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kk, device=dev) / math.sqrt(Kk)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    one = lambda: K.gemm(a, w, b, out=out)
    chunks, rem = [], N
    while rem > 0:
        c = 1 << (rem.bit_length() - 1)
        chunks.append(c); rem -= c
    def split():
        n0 = 0
        for c in chunks:
            K.gemm(a, w[n0:n0 + c], b[n0:n0 + c], out=out[:, n0:n0 + c]); n0 += c
    # alternate the two forms so clock / thermal drift hits both alike
    r1, r2 = [], []
    for _ in range(3):
        r2.append(timeit(split)); r1.append(timeit(one))
    t1, t2 = min(r1), min(r2)
    fl = 2.0 * M * N * Kk
    print(f"M={M} K={Kk} N={N:6d} (Nt={N // 256:3d}): one launch {t1:7.1f} us {fl / t1 / 1e6:7.1f} TF/s | chunks {chunks} {t2:7.1f} us {fl / t2 / 1e6:7.1f} TF/s")
