"""One 19B-size denoise loop in the opt-in fp8 COMPUTE mode (or bf16 with --bf16), eager steps, for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats -d gpurun_out/f8prof -- python tools/fp8_step.py [--bf16] [--layers 48] [--steps 8]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import VideoLatentTools
from ltx_2_mlx_amd.model.transformer import LTXModel, Modality
from ltx_2_mlx_amd.types import VideoLatentShape
ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--layers", type=int, default=48)
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = LTXModel(num_layers=a.layers, device=dev, fp8_compute=not a.bf16)
m.init_random_weights(seed=0)
g = torch.Generator(device=dev).manual_seed(3)
lat = torch.randn(3456, 128, generator=g, device=dev)
ctx = 0.1 * torch.randn(1, 1024, 3840, generator=g, device=dev)
pos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
m.prepare(ctx, pos)
sig = DISTILLED_SIGMA_VALUES
ts = torch.tensor(sig[:8], device=dev)
def run(n):
    for i in range(n):
        md = Modality(latent=lat[None], context=ctx, context_mask=None, timesteps=ts[i % 8:i % 8 + 1], positions=pos)
        m.denoise_step_(lat, md, sig[i % 8], sig[i % 8 + 1])
run(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.steps)
torch.cuda.synchronize()
print(f"{'bf16' if a.bf16 else 'fp8 compute'}: {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms/step (eager)")
