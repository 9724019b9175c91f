"""LTX-2.3-style AudioVideo DiT (BASELINE.json config 4 shape: 48 layers, video 32x128, audio 32x64,
9-row AdaLN, prompt-modulated text K/V, per-head gates) at 768x512x65: ms per joint denoise step,
eager and hipGraph, synthetic weights.  usage: python tools/bench_av.py [--v1] [--layers L] [--steps K]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, Modality
from ltx_2_mlx_amd.components import AudioPatchifier, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import AudioLatentTools, VideoLatentTools
from ltx_2_mlx_amd.types import AudioLatentShape, VideoLatentShape

ap = argparse.ArgumentParser()
ap.add_argument("--v1", action="store_true", help="19B-style AV blocks (6-row AdaLN, caption projection, cached text K/V)")
ap.add_argument("--layers", type=int, default=48)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--ab-option", default=None, help="engine option to alternate 1 / 0 over the eager loop after the main measurement (e.g. adaln_combine)")
a = ap.parse_args()
dev = torch.device("cuda:0")
v23 = not a.v1
m = LTXModel(model_type=LTXModelType.AudioVideo, num_layers=a.layers, caption_channels=None if v23 else 3840,
             cross_attention_adaln=v23, apply_gated_attention=v23, device=dev)
m.init_random_weights(seed=0)
N, Na, S = 9 * 16 * 24, 68, 1024
g = torch.Generator(device=dev).manual_seed(1)
vlat = torch.randn(N, 128, generator=g, device=dev)
alat = torch.randn(Na, 128, generator=g, device=dev)
vctx = 0.1 * torch.randn(1, S, 4096 if v23 else 3840, generator=g, device=dev)
actx = 0.1 * torch.randn(1, S, 2048 if v23 else 3840, generator=g, device=dev)
vpos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
apos = AudioLatentTools(AudioPatchifier(1), AudioLatentShape(1, 8, Na, 16)).create_initial_state(device=dev).positions
t0 = time.time()
m.prepare(vctx, vpos, audio_context=actx, audio_positions=apos)
torch.cuda.synchronize()
prep = time.time() - t0
sig = DISTILLED_SIGMA_VALUES


def eager(k):
    for i in range(k):
        s = torch.tensor([sig[i % 8]], device=dev)
        mv = Modality(latent=vlat[None], context=vctx, context_mask=None, timesteps=s, positions=vpos, sigma=s)
        ma = Modality(latent=alat[None], context=actx, context_mask=None, timesteps=s, positions=apos, sigma=s)
        m.denoise_step_(vlat, mv, sig[i % 8], sig[i % 8 + 1] or 1e-3, audio_latent=alat, audio=ma)


eager(1)
torch.cuda.synchronize()
t0 = time.time()
eager(a.steps)
torch.cuda.synchronize()
t_eager = (time.time() - t0) / a.steps
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    m.capture_denoise_graph(vlat, sig, audio_latent=alat)
    m.replay_denoise_graph()
    side.synchronize()
    t0 = time.time()
    m.replay_denoise_graph()
    side.synchronize()
    t_graph = (time.time() - t0) / 8
if a.ab_option:
    for r in range(3):
        res = []
        for v in (1, 0):
            m.set_option(a.ab_option, v)
            eager(2); torch.cuda.synchronize(); t0 = time.time(); eager(16); torch.cuda.synchronize()
            res.append((time.time() - t0) / 16 * 1e3)
        print(f"{a.ab_option}: 1 -> {res[0]:.2f} ms/step | 0 -> {res[1]:.2f} | {res[0] - res[1]:+.2f}", flush=True)
    m.set_option(a.ab_option, 1)
print(json.dumps({"workload": ("LTX-2.3" if v23 else "LTX-2 19B") + f" AudioVideo DiT {a.layers}L 768x512x65 (N=3456, Na=68, S=1024)",
                  "eager_ms_per_step": round(t_eager * 1e3, 2), "graph_ms_per_step": round(t_graph * 1e3, 2),
                  "prepare_ms": round(prep * 1e3, 1), "finite": bool(torch.isfinite(vlat).all() and torch.isfinite(alat).all())}))
