#!/usr/bin/env python
"""Headline benchmark: LTX-2 19B distilled, 768x512x65, bf16 -- denoise steps/s (+ VAE-decode frames/s).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one denoise step of the hot path over one prompt's latent: LTXModel forward
(48 blocks, D=4096, N=3456 video tokens, S=1024 text tokens) + x0 + Euler update, with inputs
and weights resident in HBM.  Each rank runs its own (prompt, seed); weights are broadcast once
from rank 0 over RCCL; there is no per-step communication (weak scaling over prompts).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def dit_algorithmic_flops(N, S, D, L):
    """per layer 8ND^2 + 4N^2D + 4ND^2 + 4SD^2 + 4NSD + 16ND^2."""
    per_layer = 8 * N * D * D + 4 * N * N * D + 4 * N * D * D + 4 * S * D * D + 4 * N * S * D + 16 * N * D * D
    return per_layer * L


def cpu_baseline(threads):
    """Oracle (fp32 PyTorch CPU port of the reference arithmetic) timed on the host cores:
    ONE full-width transformer block (N=3456, S=1024, D=4096) of one denoise step; a step is 48
    such blocks, so steps/s = 1 / (48 * t_block).  Bounded sample (about 10-30 s)."""
    from oracle import dit, loop
    torch.set_num_threads(threads)
    cfg = dit.DiTConfig(num_layers=1)
    D = cfg.inner_dim
    g = torch.Generator().manual_seed(0)
    w = {}
    for name, shape in dit.dit_weight_shapes(cfg).items():
        if name.startswith("transformer_blocks.0."):
            w[name] = torch.randn(shape, generator=g) * (0.02 if len(shape) == 2 else 1.0)
    N, S = 3456, 1024
    x = torch.randn(1, N, D, generator=g)
    ctx = torch.randn(1, S, D, generator=g) * 0.1
    emb = torch.randn(1, 1, 6, D, generator=g) * 0.1
    pe = dit.rope_split_tables(loop.video_positions(1, 9, 16, 24, 24.0), D, 32, 10000.0, [20, 2048, 2048])
    with torch.no_grad():
        t0 = time.time()
        dit.transformer_block(x, ctx, emb, pe, w, 0, cfg)
        t_block = time.time() - t0
    return {"value": 1.0 / (48 * t_block), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"1 of 48 full-width DiT blocks (N=3456,S=1024,D=4096) fp32 on {threads} host threads: "
                      f"{t_block:.2f} s/block, extrapolated x48 to one step"}


def extra_configs(dev, layers):
    """AudioVideo (LTX-2.3-style) joint step and the two-stage 1536x1024x65 pipeline, random-init weights."""
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, AudioPatchifier, VideoLatentPatchifier
    from ltx_2_mlx_amd.conditioning import AudioLatentTools, VideoLatentTools
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, decode_latent
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    from ltx_2_mlx_amd.types import AudioLatentShape, VideoLatentShape
    res = {}
    # --- config 4 shape: 48-layer AudioVideo DiT with 9-row AdaLN, prompt-modulated text K/V, per-head gates
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_layers=layers, caption_channels=None, cross_attention_adaln=True,
                 apply_gated_attention=True, device=dev)
    m.init_random_weights(seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    N, Na, S = 3456, 68, 1024
    vlat, alat = torch.randn(N, 128, generator=g, device=dev), torch.randn(Na, 128, generator=g, device=dev)
    vctx, actx = 0.1 * torch.randn(1, S, 4096, generator=g, device=dev), 0.1 * torch.randn(1, S, 2048, generator=g, device=dev)
    vpos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
    apos = AudioLatentTools(AudioPatchifier(1), AudioLatentShape(1, 8, Na, 16)).create_initial_state(device=dev).positions
    m.prepare(vctx, vpos, audio_context=actx, audio_positions=apos)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(vlat, DISTILLED_SIGMA_VALUES, audio_latent=alat)
        m.replay_denoise_graph()
        side.synchronize()
        t0 = time.perf_counter()
        m.replay_denoise_graph()
        side.synchronize()
        res["ltx23_audiovideo_ms_per_step"] = round((time.perf_counter() - t0) / 8 * 1e3, 2)
    torch.cuda.current_stream().wait_stream(side)
    del m
    torch.cuda.empty_cache()
    # --- config 5: two-stage distilled, 1536x1024x65 (8 steps @N=3456, upscaler x2, 3 steps @N=13824) + decode
    m = LTXModel(num_layers=layers, device=dev)
    m.init_random_weights(seed=0)
    dec = SimpleVideoDecoder(device=dev)
    dec.init_random_weights(seed=1)
    up = SpatialUpscaler(device=dev)
    up.init_random_weights(seed=2)
    pipe = DistilledPipeline(m, dec, None, spatial_upscaler=up)
    conf = DistilledConfig(height=1024, width=1536, num_frames=65, seed=0, use_hip_graph=True)
    ctx = 0.1 * torch.randn(1, 1024, 3840, generator=g, device=dev)
    lat = pipe(ctx, None, conf)                       # warm-up: workspaces, kernel attributes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = pipe(ctx, None, conf)
    torch.cuda.synchronize()
    res["two_stage_1536x1024x65_denoise_upscale_s"] = round(time.perf_counter() - t0, 3)
    decode_latent(lat, dec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = decode_latent(lat, dec)
    torch.cuda.synchronize()
    res["two_stage_1536x1024x65_decode_frames_per_sec"] = round(fr.shape[0] / (time.perf_counter() - t0), 1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=48, help="debug only; the headline config is 48")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay section (rocprofv3 --pmc passes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (AudioVideo step, two-stage pipeline)")
    args = ap.parse_args()

    from ltx_2_mlx_amd import _native as nv
    from ltx_2_mlx_amd import distributed as D
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
    from ltx_2_mlx_amd.conditioning import VideoLatentTools
    from ltx_2_mlx_amd.components import VideoLatentPatchifier
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, decode_latent
    from ltx_2_mlx_amd.types import VideoLatentShape, VideoPixelShape

    nv.lib()    # fail loudly if the HIP extension is missing
    rank, world, local = D.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # ---------------- model (random init of the 19B architecture; rank 0 -> RCCL broadcast) ----------------
    L = args.layers
    model = LTXModel(num_layers=L, device=dev)
    model.init_random_weights(seed=0 if rank == 0 else 1000 + rank)
    t0 = time.time()
    n_coll = D.broadcast_tensors(model.weight_tensors(), src=0, bucket_bytes=1 << 30)
    torch.cuda.synchronize()
    bcast_s = time.time() - t0

    # ---------------- per-rank prompt / seed ----------------
    shape = VideoLatentShape.from_pixel_shape(VideoPixelShape(1, 65, 512, 768, 24.0))
    N = shape.frames * shape.height * shape.width
    S, Dm = 1024, model.inner_dim
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    tools = VideoLatentTools(VideoLatentPatchifier(1), shape, fps=24.0)
    state = tools.create_initial_state(device=dev)
    noise = torch.randn(N, 128, generator=g, device=dev)
    ctx = 0.1 * torch.randn(1, S, 3840, generator=g, device=dev)
    t0 = time.time()
    model.prepare(ctx, state.positions)
    torch.cuda.synchronize()
    prep_ms = (time.time() - t0) * 1e3

    sig = DISTILLED_SIGMA_VALUES
    K, W = args.steps, args.warmup
    lat = noise.clone()

    def run_steps(n, profile=False):
        for i in range(n):
            s0, s1 = sig[i % 8], sig[i % 8 + 1]
            if i % 8 == 0:
                lat.copy_(noise)
            m = Modality(latent=lat[None], context=ctx, context_mask=None, timesteps=ts_dev[i % 8:i % 8 + 1], positions=state.positions)
            model.denoise_step_(lat, m, s0, s1)

    ts_dev = torch.tensor(sig[:8], device=dev)
    run_steps(W)
    torch.cuda.synchronize()

    # ---------------- timed region: exactly K steps, dominant GEMM bracketed by HIP events ----------------
    DOM_EPI = nv.EPI_RESID_GATE_F32     # gemm_pp_kernel<4,false,224>: attn1.to_out, attn2.to_out, ff.net.2 (+ gated residual)
    D.barrier()
    torch.cuda.synchronize()
    model.profile_begin(DOM_EPI)
    t0 = time.perf_counter()
    run_steps(K)
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    k_ms, k_n, k_fl = model.profile_end()
    dt = D.max_over_ranks(dt, dev)

    # ---------------- hipGraph replay of the 8-step loop (reported beside the headline) ----------------
    graph_ms = None
    try:
        if args.no_graph:
            raise RuntimeError("skipped (--no-graph)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            lat.copy_(noise)
            model.capture_denoise_graph(lat, sig)
            model.replay_denoise_graph()
            side.synchronize()
            reps = max(1, K // 8)
            t0 = time.perf_counter()
            for _ in range(reps):
                model.replay_denoise_graph()
            side.synchronize()
            graph_ms = (time.perf_counter() - t0) / (reps * 8) * 1e3
        torch.cuda.current_stream().wait_stream(side)
    except Exception as e:  # noqa: BLE001
        graph_ms = f"failed: {e}"

    # ---------------- VAE decode: latent in HBM -> uint8 frames in HBM ----------------
    vae_fps, vae_ms = None, None
    if not args.no_vae:
        dec = SimpleVideoDecoder(device=dev)
        dec.init_random_weights(seed=7)
        dec.generator = torch.Generator(device=dev).manual_seed(99 + rank)
        z = torch.randn(1, 128, shape.frames, shape.height, shape.width, generator=g, device=dev)
        frames = decode_latent(z, dec)                    # warm-up (also sizes the workspace)
        torch.cuda.synchronize()
        D.barrier()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            frames = decode_latent(z, dec)
        torch.cuda.synchronize()
        D.barrier()
        vdt = D.max_over_ranks((time.perf_counter() - t0) / reps, dev)
        assert tuple(frames.shape) == (65, 512, 768, 3)
        vae_ms = vdt * 1e3
        vae_fps = world * 65 / vdt

    if rank != 0:
        return
    steps_per_s = world * K / dt
    ms_per_step = dt / K * 1e3
    alg = dit_algorithmic_flops(N, S, Dm, L)
    # HBM/fabric traffic of the dominant kernel cannot be collected from inside the process; it comes
    # from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json), null if absent.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            traffic = json.load(f).get("per_launch_avg_bytes")
    except Exception:  # noqa: BLE001
        pass
    kern_avg_ms = k_ms / max(k_n, 1)
    kern_tflops = (k_fl / max(k_n, 1)) / (kern_avg_ms * 1e-3) / 1e12
    out = {
        "metric": "denoise_steps_per_sec", "value": round(steps_per_s, 4), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"LTX-2 19B distilled DiT ({L} layers, D=4096, 32x128 heads), 768x512x65 "
                               f"(N={N} video tokens, S={S} text tokens), 8-step distilled sigmas, bf16, random-init weights",
                   "parallelism": f"prompt-parallel x{world} (independent prompt/seed per GPU, one RCCL weight broadcast)"},
        "vae_decode_frames_per_sec": None if vae_fps is None else round(vae_fps, 2),
        "vae_decode_ms": None if vae_ms is None else round(vae_ms, 2),
        "step_algorithmic_tflop": round(alg / 1e12, 3),
        "step_mfma_roofline_frac": round(alg / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "hipgraph_ms_per_step": graph_ms if not isinstance(graph_ms, float) else round(graph_ms, 3),
        "prompt_setup_ms": round(prep_ms, 1), "weight_broadcast_s": round(bcast_s, 3), "weight_broadcast_collectives": n_coll,
        "roofline": {"kernel": "gemm_pp_kernel<4,false,224> (224x256x64 ping-pong bf16 MFMA GEMM, EPI_RESID_GATE_F32: attn1/attn2 to_out and ff.net.2 + bias + gate*residual)",
                     "bound": "mfma", "achieved": round(kern_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(kern_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                     "launches": k_n, "avg_launch_us": round(kern_avg_ms * 1e3, 2),
                     "algorithmic_flops_per_launch": round(k_fl / max(k_n, 1))},
    }
    if world == 1 and not args.no_extra:
        # Secondary BASELINE configurations, reported beside the headline (never part of `value`): config 4 shape
        # (LTX-2.3-style AudioVideo DiT, joint audio+video step) and config 5 (two-stage 1536x1024x65 with the
        # spatial upscaler).  Any failure here is reported as text and cannot affect the numbers above.
        try:
            del model
            torch.cuda.empty_cache()
            out["extra_configs"] = extra_configs(dev, L)
        except Exception as e:  # noqa: BLE001
            out["extra_configs"] = {"error": str(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
