"""Denoise step time with PER-TOKEN timesteps (image conditioning: timesteps = denoise_mask * sigma, per-row AdaLN gates) beside the uniform-sigma
step, 48 layers at the BASELINE geometry, eager."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import VideoLatentTools
from ltx_2_mlx_amd.model.transformer import LTXModel, Modality
from ltx_2_mlx_amd.types import VideoLatentShape
dev = torch.device("cuda:0")
m = LTXModel(num_layers=48, device=dev)
m.init_random_weights(seed=0)
g = torch.Generator(device=dev).manual_seed(3)
lat = torch.randn(3456, 128, generator=g, device=dev)
ctx = 0.1 * torch.randn(1, 1024, 3840, generator=g, device=dev)
pos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
sig = DISTILLED_SIGMA_VALUES
mask = torch.ones(3456, device=dev)
mask[:384] = 0.0                                   # first latent frame conditioned
for name, per_token in (("uniform", False), ("per-token", True)):
    def run(n):
        for i in range(n):
            s = sig[i % 8]
            ts = (mask * s).reshape(1, -1, 1) if per_token else torch.tensor([s], device=dev)
            md = Modality(latent=lat[None], context=ctx, context_mask=None, timesteps=ts, positions=pos)
            m.denoise_step_(lat, md, sig[i % 8], sig[i % 8 + 1], denoise_mask=mask[:, None].expand(-1, 128).contiguous() if per_token else None,
                            clean_latent=lat.clone() if per_token else None)
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(8)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step (eager)")
