"""Round 6 A/B inside one process: the 48-layer bf16 step with the text cross-attention's pre-norm folded around its GEMMs (engine option fold_norms = 1, the default)
against round 5's form (0: a norm pass), alternating, 16 steps each.  Prints ms/step and the latent's distance after 8 steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.model.transformer import LTXModel, Modality
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import VideoLatentTools
from ltx_2_mlx_amd.types import VideoLatentShape
dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 48
m = LTXModel(num_layers=layers, device=dev)
m.init_random_weights(seed=0)
shape = VideoLatentShape(1, 128, 9, 16, 24)
N = shape.frames * shape.height * shape.width
g = torch.Generator(device=dev).manual_seed(1)
state = VideoLatentTools(VideoLatentPatchifier(1), shape, fps=24.0).create_initial_state(device=dev)
noise = torch.randn(N, 128, generator=g, device=dev)
ctx = 0.1 * torch.randn(1, 1024, 3840, generator=g, device=dev)
m.prepare(ctx, state.positions)
sig = DISTILLED_SIGMA_VALUES
ts = torch.tensor(sig[:8], device=dev)
lat = noise.clone()
def steps(n):
    for i in range(n):
        if i % 8 == 0: lat.copy_(noise)
        m.denoise_step_(lat, Modality(latent=lat[None], context=ctx, context_mask=None, timesteps=ts[i % 8:i % 8 + 1], positions=state.positions), sig[i % 8], sig[i % 8 + 1])
def timed(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); steps(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
outs = {}
for lv in (0, 1):
    m.set_option("fold_norms", lv); steps(8); outs[lv] = lat.clone()
rl = lambda a, b: float((a - b).double().norm() / b.double().norm())
print(f"8-step latent: folded vs unfolded rel-L2 {rl(outs[1], outs[0]):.3e}", flush=True)
for r in range(6):
    t = {}
    for lv in ((1, 0) if r % 2 else (0, 1)):
        m.set_option("fold_norms", lv); steps(8); t[lv] = timed(16)
    print(f"unfolded {t[0]:.3f} | folded {t[1]:.3f} ({t[1] - t[0]:+.3f}) ms/step", flush=True)
