"""The two attention kernel forms (32 / 64 query rows per wave) side by side on the DiT's self-attention shapes (same box, alternating, best of 3),
with the difference between them and against fp64 on a sample of rows, and the 64-row form's cost attribution (its timing-experiment builds:
no exponentials / no row maximum -- those results are wrong by construction).  Numbers of record: profiles/r04_attn_64row_negative.md."""
import math, os, sys, torch
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
FORMS = {1: "32 rows per wave (production)", 2: "64 rows per wave", 3: "64-row, no exponentials", 18: "64-row, no row maximum / rescale", 19: "64-row, neither"}
for (Nq, Nkv, H) in [(3456, 3456, 32), (13824, 13824, 32), (3456, 100, 32), (300, 3456, 8), (1000, 50, 2)]:
    hd = 128
    D = H * hd
    g = torch.Generator(device=dev).manual_seed(Nq + Nkv)
    q = torch.randn(Nq, D, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Nkv, D, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Nkv, D, device=dev, generator=g).to(torch.bfloat16)
    vt = K.vt_transpose(v, H, head_dim=hd)
    a = K.flash_attn_form(q, k, vt, H, Nkv, 1)
    b = K.flash_attn_form(q, k, vt, H, Nkv, 2)
    diff = (a.float() - b.float()).abs().max().item()
    rows = torch.arange(0, Nq, max(1, Nq // 97), device=dev)
    qh, kh, vh = [t.reshape(-1, H, hd).transpose(0, 1) for t in (q[rows].double(), k.double(), v.double())]
    ex = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh).transpose(0, 1).reshape(len(rows), D)
    ea = ((a[rows].double() - ex).norm() / ex.norm()).item(); eb = ((b[rows].double() - ex).norm() / ex.norm()).item()
    print(f"Nq={Nq} Nkv={Nkv} H={H}: max |32-row - 64-row| {diff:.3g} | rel-L2 vs fp64: {ea:.2e} / {eb:.2e}", flush=True)
    fl = 4.0 * Nq * Nkv * D
    for form, name in FORMS.items():
        best = min(timeit(lambda: K.flash_attn_form(q, k, vt, H, Nkv, form)) for _ in range(3))
        print(f"    {name:36s} {best*1e6:8.1f} us {fl/best/1e12:7.1f} TF/s", flush=True)
