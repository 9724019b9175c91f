"""Same-box A/B of an engine option on the headline step: the captured 8-step loop of the 48-layer 19B model (768x512x65) replayed with the
option on and off, alternating.  usage: python tools/qk_fold_ab.py [option=qk_fold] [rounds=3] [layers=48]"""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier
from ltx_2_mlx_amd.conditioning import VideoLatentTools
from ltx_2_mlx_amd.model.transformer import LTXModel
from ltx_2_mlx_amd.types import VideoLatentShape

opt = sys.argv[1] if len(sys.argv) > 1 else "qk_fold"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 48
dev = torch.device("cuda:0")
m = LTXModel(num_layers=layers, device=dev)
m.init_random_weights(seed=0)
g = torch.Generator(device=dev).manual_seed(3)
lat = torch.randn(3456, 128, generator=g, device=dev)
ctx = 0.1 * torch.randn(1, 1024, 3840, generator=g, device=dev)
pos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
side = torch.cuda.Stream()
vals = [int(v) for v in os.environ.get("AB_VALUES", "0,1").split(",")]
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        m.set_option(opt, v)
        m.prepare(ctx, pos)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            z = lat.clone()
            m.capture_denoise_graph(z, DISTILLED_SIGMA_VALUES)
            m.replay_denoise_graph()
            side.synchronize()
            runs = []
            for _ in range(3):
                t0 = time.perf_counter()
                m.replay_denoise_graph()
                side.synchronize()
                runs.append((time.perf_counter() - t0) / 8 * 1e3)
        torch.cuda.current_stream().wait_stream(side)
        res[v].append(statistics.median(runs))
        print(f"round {r} {opt}={v}: {res[v][-1]:.3f} ms/step {[round(x, 3) for x in runs]}", flush=True)
print(" | ".join(f"{opt}={v}: best {min(res[v]):.3f} median {statistics.median(res[v]):.3f}" for v in vals), "ms/step")
