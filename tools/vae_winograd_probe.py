"""Round 6: would Winograd F(2x2, 3x3) over (H, W) per temporal tap hold the VAE decoder's accuracy with 16-bit MFMA operands?
A numerics EMULATION, checker-side only (torch fp32 ops on the GPU; nothing here is on the product path): the fp32 oracle decoder (oracle/vae.py) is run three times on
the same latent and weights with its conv3d replaced by
  fp32      the oracle's own convolution
  direct    what the HIP kernels compute: inputs and weights rounded to bfloat16, products accumulated in fp32
  winograd  the res-block convs of the 128- and 256-channel stages (63 % of the decode's flops) as F(2x2, 3x3): U = G g G^T and V = B^T d B formed in fp32 and ROUNDED TO
            bfloat16 (they are the MFMA operands), the 16 element-wise channel contractions and the three temporal taps accumulated in fp32, Y = A^T M A in fp32;
            every other conv as `direct`
and the float video of each is compared with the fp32 run (the test suite's gate on that figure is 3e-2, `direct` measures ~7.5e-3 at full size).
Also prints, per conv shape, the single-conv error of both forms against fp64.  usage: python tools/vae_winograd_probe.py [--f16]"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae
dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
LP = torch.float16 if "--f16" in sys.argv else torch.bfloat16
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)
q = lambda t: t.to(LP).to(t.dtype)


def pad(x, causal, k=3):
    p = (k - 1) // 2
    x = torch.cat([x[:, :, :, 1:p + 1].flip(3), x, x[:, :, :, -(p + 1):-1].flip(3)], dim=3)
    x = torch.cat([x[..., 1:p + 1].flip(4), x, x[..., -(p + 1):-1].flip(4)], dim=4)
    tp = k - 1
    if causal:
        return torch.cat([x[:, :, :1].repeat(1, 1, tp, 1, 1), x], dim=2)
    return torch.cat([x[:, :, :1].repeat(1, 1, 1, 1, 1), x, x[:, :, -1:].repeat(1, 1, 1, 1, 1)], dim=2)


def winograd(xp, w, bias):
    """xp [1, C, T+2, H+2, W+2] padded (values already 16-bit), w [O, C, 3, 3, 3] -> [1, O, T, H, W]; H, W even."""
    Bn, C, Tp, Hp, Wp = xp.shape
    O = w.shape[0]
    T, H, W = Tp - 2, Hp - 2, Wp - 2
    U = q(torch.einsum("pi,ocdij,qj->ocdpq", G, w.float(), G))                       # [O, C, 3, 4, 4], rounded: an MFMA operand
    out = torch.zeros(Bn, O, T, H // 2, W // 2, 4, 4, device=xp.device)
    for t0 in range(0, T, 4):                                                        # frames in slabs: V is 4x the activation
        t1 = min(T, t0 + 4)
        d = xp[:, :, t0:t1 + 2].unfold(3, 4, 2).unfold(4, 4, 2)                       # [1, C, t+2, H/2, W/2, 4, 4]
        V = q(torch.einsum("pi,nctxyij,qj->nctxypq", BT, d, BT))                     # rounded: the other MFMA operand
        for kt in range(3):
            out[:, :, t0:t1] += torch.einsum("ocpq,nctxypq->notxypq", U[:, :, kt], V[:, :, kt:kt + (t1 - t0)])
    Y = torch.einsum("ip,notxypq,jq->notxyij", AT, out, AT)                          # [1, O, T, H/2, W/2, 2, 2]
    Y = Y.permute(0, 1, 2, 3, 5, 4, 6).reshape(Bn, O, T, H, W)
    return Y + bias.float()[None, :, None, None, None]


MODE = ["fp32"]


def conv_emul(x, weight, bias, causal=False):
    if MODE[0] == "fp32" or weight.shape[2] != 3:
        return ORIG(x, weight, bias, causal)
    xp = pad(q(x.float()), causal)
    O, C = weight.shape[:2]
    if MODE[0] == "winograd" and O == C and C in (128, 256) and xp.shape[3] % 2 == 0 and xp.shape[4] % 2 == 0:
        return winograd(xp, weight, bias)
    return F.conv3d(xp, q(weight.float()), bias.float())


ORIG = vae.conv3d_simple
vae.conv3d_simple = conv_emul
torch.manual_seed(0)
# ---- single convs against fp64
SMALL = dev.type == "cpu"          # (a container without a GPU: tiny shapes, to check the algebra only)
for C, (T, H, W) in (((128, (2, 8, 8)),) if SMALL else ((128, (5, 64, 96)), (256, (5, 32, 48)))):
    x = F.silu(torch.randn(1, C, T, H, W, device=dev))
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.02
    b = torch.zeros(C, device=dev)
    xp = pad(q(x), False)
    ref = F.conv3d(pad(x, False).double(), w.double())
    e = lambda y: float((y.double() - ref).norm() / ref.norm())
    print(f"single conv {C}->{C}: direct {LP} rel-L2 {e(F.conv3d(xp, q(w))):.3e} | winograd F(2x2,3x3) {e(winograd(xp, w, b)):.3e}", flush=True)
# ---- the whole decoder
if SMALL:
    sys.exit(0)
cfg = vae.VAEConfig()
wts = {k: v.to(dev) for k, v in vae.make_vae_weights(cfg, seed=5).items()}
lat = torch.randn(1, 128, 3, 6, 8, device=dev)
outs = {}
with torch.device(dev), torch.no_grad():
    for m in ("fp32", "direct", "winograd"):
        MODE[0] = m
        outs[m] = vae.decoder_forward(lat, wts, cfg, timestep=0.05)
rl = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
u8 = lambda v: (((v + 1) / 2).clamp(0, 1) * 255).floor()
print(f"decoder {tuple(outs['fp32'].shape)}: float video rel-L2 vs fp32: direct {rl(outs['direct'], outs['fp32']):.3e} | winograd {rl(outs['winograd'], outs['fp32']):.3e}  (gate 3e-2)")
print(f"uint8 frames mean |diff| vs fp32: direct {float((u8(outs['direct']) - u8(outs['fp32'])).abs().mean()):.3f} | winograd {float((u8(outs['winograd']) - u8(outs['fp32'])).abs().mean()):.3f}")
