"""Round 4 A/B inside one process: the 48-layer step with the AdaLN rows of every layer combined once per step (engine default) against the
round-3 form (option adaln_combine = 0: tables and embeddings reach every kernel separately), alternating, 16 steps each.  Bit-identical outputs."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.model.transformer import LTXModel, Modality
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import VideoLatentTools
from ltx_2_mlx_amd.types import VideoLatentShape
dev = torch.device("cuda:0")
m = LTXModel(num_layers=48, device=dev)
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
steps(8); out1 = lat.clone()
m.set_option("adaln_combine", 0); steps(8); out0 = lat.clone()
print("bit-identical:", bool(torch.equal(out0, out1)))
for r in range(4):
    m.set_option("adaln_combine", 1); steps(4); a = timed(16)
    m.set_option("adaln_combine", 0); steps(4); b = timed(16)
    print(f"combined {a:.3f} ms/step | separate {b:.3f} ms/step | {a - b:+.3f}", flush=True)
