"""Socket power while ONE kernel runs back to back for a few seconds (rocm-smi polled from a thread): is it at the 1400 W cap?
usage: python tools/kernel_power.py attn|attn_cross|qkv|cross_q|to_out|ffn_up|ffn_down|norm|vae   (prints pJ per useful flop = W / (TF/s))"""
import math, os, re, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltx_2_mlx_amd.kernels as K
from ltx_2_mlx_amd import _native as nv
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "attn"
N, D, H = 3456, 4096, 32
if which.startswith("attn"):
    nkv = 1024 if which == "attn_cross" else N
    q = torch.randn(N, D, device=dev).to(torch.bfloat16); k = torch.randn(nkv, D, device=dev).to(torch.bfloat16)
    vt = K.vt_transpose(torch.randn(nkv, D, device=dev).to(torch.bfloat16), H)
    fn = lambda: K.flash_attn(q, k, vt, H, nkv)
    flop = 4.0 * N * nkv * D
elif which == "vae":
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, decode_latent
    dec = SimpleVideoDecoder(device=dev)
    dec.init_random_weights(seed=7)
    dec.generator = torch.Generator(device=dev).manual_seed(99)
    z = torch.randn(1, 128, 9, 16, 24, device=dev)
    fn = lambda: decode_latent(z, dec)
    flop = 37.7e12
elif which in ("gemm", "ffn_up", "qkv", "cross_q", "to_out", "ffn_down", "gemm_resid"):
    # the DiT layer's projections at N = 3456, D = 4096: qkv (N = 3D), cross_q (N = D), ffn_up (N = 4D, GELU) write 16 bits; to_out (K = D) and
    # ffn_down (K = 4D) add gate * (.) into the fp32 residual ("gemm" = ffn_up without GELU, "gemm_resid" = to_out: round-4 names)
    nout, kin, epi = {"gemm": (4 * D, D, nv.EPI_BF16), "ffn_up": (4 * D, D, nv.EPI_GELU_BF16), "qkv": (3 * D, D, nv.EPI_BF16), "cross_q": (D, D, nv.EPI_BF16),
                      "to_out": (D, D, nv.EPI_RESID_GATE_F32), "gemm_resid": (D, D, nv.EPI_RESID_GATE_F32), "ffn_down": (D, 4 * D, nv.EPI_RESID_GATE_F32)}[which]
    a = torch.randn(N, kin, device=dev).to(torch.bfloat16); w = (torch.randn(nout, kin, device=dev) / 64).to(torch.bfloat16)
    bias = 0.02 * torch.randn(nout, device=dev)
    if epi == nv.EPI_RESID_GATE_F32:
        out = torch.zeros(N, nout, device=dev); gt = 0.01 * torch.randn(nout, device=dev)         # (small gate: the residual stays finite over thousands of launches)
        fn = lambda: K.gemm(a, w, bias, epilogue=epi, out=out, gate_table=gt)
    else:
        out = torch.empty(N, nout, device=dev, dtype=torch.bfloat16)
        fn = lambda: K.gemm(a, w, bias, epilogue=epi, out=out)
    flop = 2.0 * N * nout * kin
else:
    x = torch.randn(N, D, device=dev); t = [0.1 * torch.randn(D, device=dev) for _ in range(4)]
    fn = lambda: K.adaln_rmsnorm(x, 1e-6, False, *t, 0)
    flop = 0.0
samples, stop = [], False
def poll():
    while not stop:
        o = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
        for l in o.splitlines():
            if l.startswith("card0"):
                f = l.split(",")
                samples.append((float(f[-1]), int(re.sub(r"\D", "", f[6]))))
th = threading.Thread(target=poll); th.start()
reps = 4 if which == "vae" else 200
for _ in range(2 if which == "vae" else 20): fn()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(reps): fn()
    torch.cuda.synchronize(); n += reps
dt = time.time() - t0
stop = True; th.join()
busy = [s for s in samples[len(samples) // 4:]]
pj = (sum(s[0] for s in busy) / len(busy)) / (flop / (dt / n) / 1e12) if flop else float("nan")
print(f"{which}: {dt / n * 1e6:.1f} us per launch, {flop / (dt / n) / 1e12:.0f} TF/s, {pj:.2f} pJ per useful flop; socket power mean {sum(s[0] for s in busy) / len(busy):.0f} W (max {max(s[0] for s in busy):.0f}), sclk mean {sum(s[1] for s in busy) / len(busy):.0f} MHz over {len(busy)} samples")
