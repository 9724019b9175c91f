#!/usr/bin/env python
"""Headline benchmark: LTX-2 19B distilled, 768x512x65, bf16 -- denoise steps/s (+ VAE-decode frames/s).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script re-launches itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); under a
launcher (RANK / WORLD_SIZE set) it runs as the rank it is.

A "step" is one denoise step of the hot path over one prompt's latent: LTXModel forward
(48 blocks, D=4096, N=3456 video tokens, S=1024 text tokens) + x0 + Euler update, with inputs
and weights resident in HBM.  Each rank runs its own (prompt, seed); weights are broadcast once
from rank 0 over RCCL; there is no per-step communication (weak scaling over prompts).
The headline K steps are timed with NO instrumentation; the dominant GEMM's launch time comes from a second,
untimed pass of K steps with HIP events around every launch of that kernel.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
# one 768x512x65 decode with the reference's 7/2 chunking
VAE_ALG_TFLOP, VAE_ALG_GB = 37.73, 17.2
CPU_THREADS = 32               # the oracle's fp32 block is fastest around 32 threads on the boxes' 2 x 64-core hosts
                               # (measured s/block: 32 thr 2.4, 64 thr 2.8, 128 thr 4.5, 256 thr 13.9)


def dit_algorithmic_flops(N, S, D, L):
    """per layer 8ND^2 + 4N^2D + 4ND^2 + 4SD^2 + 4NSD + 16ND^2."""
    per_layer = 8 * N * D * D + 4 * N * N * D + 4 * N * D * D + 4 * S * D * D + 4 * N * S * D + 16 * N * D * D
    return per_layer * L


def dit_executed_flops(N, S, D, L):
    """What the engine EXECUTES per step: the text cross-attention K / V projections (4 S D^2 per layer in the algorithmic count; the reference
    recomputes them every step, model.py:262-271) are step-invariant for the 19B model and run once per prompt in ltx2_dit_prepare."""
    return dit_algorithmic_flops(N, S, D, L) - 4 * S * D * D * L


def kernel_source_sha():
    """sha256 of the dominant kernel's sources (tools/pmc_traffic.py stores the same digest beside the PMC traffic figure)"""
    import hashlib
    root = os.path.join(ROOT, "ltx-2-mlx_amd", "csrc")
    h = hashlib.sha256()
    for f in ("gemm_v4.hip", "gemm_v4_loop.inc", "gemm_epilogue.h", "gemm.h", "common.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(threads):
    """The oracle (fp32 PyTorch CPU port of the reference's arithmetic, kind "port") timed on the host cores, as SURVEY.md
    8(d) defines the CPU leg -- bounded to about half a minute:
      (i)   DiT: ONE denoise step of an L=2 model at full width (D=4096, N=3456, S=1024), median of 3 after one warm-up,
            x24 linear extrapolation to the 48-layer step (labelled);
      (ii)  the plumbing config un-extrapolated: 256x384x17 (N=288), S=256, 2-layer DiT, the whole 8-step distilled loop;
      (iii) VAE: decode of ONE 2-latent-frame chunk (1,128,2,16,24) -> 9 frames at 512x768 with the default decoder
            (base_channels 128), scaled by algorithmic FLOPs to the chunked 65-frame decode (labelled)."""
    from oracle import dit, loop, vae
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    out = {"unit": "steps/s", "cores": threads, "kind": "port"}
    with torch.no_grad():
        # (i) full-width L=2 step
        cfg = dit.DiTConfig(num_layers=2)
        w = dit.make_dit_weights(cfg, seed=0)
        N, S = 3456, 1024
        lat = torch.randn(1, N, 128, generator=g)
        ctx = 0.1 * torch.randn(1, S, cfg.caption_channels, generator=g)
        pos = loop.video_positions(1, 9, 16, 24, 24.0)
        ts = torch.tensor([1.0])
        runs = []
        for i in range(4):
            t0 = time.time()
            dit.x0_model(lat, ctx, ts, pos, w, cfg)
            runs.append(time.time() - t0)
        t_l2 = statistics.median(runs[1:])
        del w
        # (ii) plumbing config, whole loop
        lat5 = torch.randn(1, 128, 3, 8, 12, generator=g)
        ctx2 = 0.1 * torch.randn(1, 256, cfg.caption_channels, generator=g)
        w2 = dit.make_dit_weights(cfg, seed=1)
        pos2 = loop.video_positions(1, 3, 8, 12, 24.0)
        t0 = time.time()
        loop.denoise_loop_cli(lat5, lambda tok, s: dit.x0_model(tok, ctx2, torch.tensor([s]), pos2, w2, cfg), loop.DISTILLED_SIGMA_VALUES)
        t_plumb = time.time() - t0
        del w2
        # (iii) one VAE chunk
        vcfg = vae.VAEConfig()
        vw = vae.make_vae_weights(vcfg, seed=2)
        z = torch.randn(1, 128, 2, 16, 24, generator=g)
        t0 = time.time()
        vid = vae.decoder_forward(z, vw, vcfg, timestep=0.05)
        t_vae = time.time() - t0
        frames = vid.shape[2]
    # algorithmic FLOPs of a T'-latent-frame decoder pass scale with the output frame count 8T'-7 (32.65 TF at 65)
    chunk_tf = 32.65 * frames / 65.0
    t_decode = t_vae * VAE_ALG_TFLOP / chunk_tf
    out.update({
        "value": 1.0 / (24 * t_l2),
        "sample": f"one denoise step of a 2-layer full-width DiT (D=4096, N=3456, S=1024) in fp32 on {threads} host threads: "
                  f"median of 3 after a warm-up {t_l2:.2f} s (runs {[round(r, 2) for r in runs]}), x24 -> 48 layers",
        "dit_l2_step_s": round(t_l2, 3),
        "plumbing_config_8_steps_s": round(t_plumb, 3),
        "plumbing_config": "256x384x17 (N=288), S=256, 2-layer DiT, 8 distilled steps, un-extrapolated",
        "vae_chunk_s": round(t_vae, 2), "vae_chunk_frames": int(frames),
        "vae_decode_frames_per_sec": round(65.0 / t_decode, 3),
        "vae_sample": f"decoder pass of one (1,128,2,16,24) chunk -> {frames} frames at 512x768 (base_channels 128) in {t_vae:.1f} s, "
                      f"scaled by algorithmic FLOPs ({chunk_tf:.2f} of {VAE_ALG_TFLOP} TF) to the 7/2-chunked 65-frame decode",
    })
    return out


def _median_replay_ms(model, side, steps=8, reps=3):
    """Median over `reps` timed replays of the captured `steps`-step graph (after one untimed replay), ms per step."""
    model.replay_denoise_graph()
    side.synchronize()
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        model.replay_denoise_graph()
        side.synchronize()
        runs.append((time.perf_counter() - t0) / steps * 1e3)
    return round(statistics.median(runs), 2), [round(r, 2) for r in runs]


def _median_s(fn, reps=3):
    fn()                                                  # warm-up: workspaces, kernel attributes
    torch.cuda.synchronize()
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
    return statistics.median(runs), runs, out


def loader_throughput(dev, layers):
    """load_transformer_weights on a synthetic safetensors checkpoint of `layers` 19B-width blocks (bf16, the reference's key
    scheme), written to the box's scratch disk first: file -> pinned staging -> HBM GB/s (loader/weight_converter.py
    SafetensorsStream).  The second load reads the file from the page cache; both are reported."""
    import tempfile
    from safetensors.torch import save_file
    from ltx_2_mlx_amd.loader import load_transformer_weights
    from ltx_2_mlx_amd.model.transformer import LTXModel
    m = LTXModel(num_layers=layers, device=dev)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in m.expected_weight_shapes().items():
        big = len(shp) == 2 and shp[0] * shp[1] >= 1 << 20
        # one random block tiled over the big matrices: host RNG for 19 G parameters would take minutes and says nothing about the loader
        if big:
            blk = (0.02 * torch.randn(256, shp[1], generator=g)).to(torch.bfloat16)
            t = blk.repeat((shp[0] + 255) // 256, 1)[:shp[0]].contiguous()
        else:
            t = (0.02 * torch.randn(*shp, generator=g)).to(torch.bfloat16 if len(shp) == 2 else torch.float32)
        sd["model.diffusion_model." + k] = t
    d = tempfile.mkdtemp(prefix="ltx2_loader_")
    path = os.path.join(d, "synthetic.safetensors")
    t0 = time.perf_counter()
    save_file(sd, path, metadata={"model_version": "2.0.0"})
    t_write = time.perf_counter() - t0
    del sd
    size = os.path.getsize(path)
    res = {"file_gb": round(size / 1e9, 2), "layers": layers, "write_s": round(t_write, 1)}
    try:
        for tag in ("first_load", "cached_load"):
            t0 = time.perf_counter()
            st = load_transformer_weights(m, path, strict=True)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            res[tag] = {"file_to_hbm_gbps": round(st["bytes"] / st["seconds"] / 1e9, 2), "file_to_hbm_s": round(st["seconds"], 2),
                        "incl_weight_packing_s": round(wall, 2)}
    finally:
        os.remove(path)
        os.rmdir(d)
    return res


def extra_configs(dev, layers):
    """Secondary BASELINE configurations, random-init weights, every timing the MEDIAN of 3 after a warm-up: config 3 (fp8-resident
    weights), config 4 (LTX-2.3 AudioVideo joint step), config 5 (two-stage 1536x1024x65: denoise + upscale, whole-volume decode
    and the TILED decode the pipeline itself takes above 4000 latent voxels)."""
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, AudioPatchifier, VideoLatentPatchifier
    from ltx_2_mlx_amd.conditioning import AudioLatentTools, VideoLatentTools
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, TilingConfig, decode_latent, decode_tiled
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    from ltx_2_mlx_amd.types import AudioLatentShape, VideoLatentShape
    res = {"timing": "median of 3 timed repetitions after one warm-up; the individual runs are listed beside each figure"}
    # --- config 3: the same 19B step with fp8-RESIDENT weights (e4m3fn codes + scale in HBM, expanded inside the GEMM; outputs
    #     bit-identical to dequantising at load -- tests/test_parity_fullsize.py)
    from ltx_2_mlx_amd.conditioning import VideoLatentTools as _VLT
    m = LTXModel(num_layers=layers, device=dev)
    m.init_random_weights(seed=0, fp8_resident=True)
    g3 = torch.Generator(device=dev).manual_seed(3)
    lat3 = torch.randn(3456, 128, generator=g3, device=dev)
    ctx3 = 0.1 * torch.randn(1, 1024, 3840, generator=g3, device=dev)
    pos3 = _VLT(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=24.0).create_initial_state(device=dev).positions
    m.prepare(ctx3, pos3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(lat3, DISTILLED_SIGMA_VALUES)
        res["fp8_resident_ms_per_step"], res["fp8_resident_runs"] = _median_replay_ms(m, side)
        res["fp8_resident_weight_gb"] = round(sum(t.numel() * t.element_size() for t in m.weight_tensors().values()) / 1e9, 2)
    torch.cuda.current_stream().wait_stream(side)
    del m
    torch.cuda.empty_cache()
    # --- config 3 as BASELINE names it ("fp8 weights (CDNA4 fp8 MFMA)"): opt-in fp8 COMPUTE -- e4m3fn weights x per-token e4m3fn activations
    #     on v_mfma_f32_32x32x64_f8f6f4 in the six projections of every block; accuracy against the fp32 oracle: tests/test_parity_fullsize.py
    m = LTXModel(num_layers=layers, device=dev, fp8_compute=True)
    m.init_random_weights(seed=0)
    m.prepare(ctx3, pos3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(lat3, DISTILLED_SIGMA_VALUES)
        res["fp8_compute_ms_per_step"], res["fp8_compute_runs"] = _median_replay_ms(m, side)
    torch.cuda.current_stream().wait_stream(side)
    del m
    torch.cuda.empty_cache()
    # --- the headline step at the reference's DEFAULT precision (scripts/generate.py:1006 computes in float16): libltx2hip_f16.so, same geometry
    m = LTXModel(num_layers=layers, device=dev, compute_dtype=torch.float16)
    m.init_random_weights(seed=0)
    m.prepare(ctx3, pos3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(lat3, DISTILLED_SIGMA_VALUES)
        res["fp16_ms_per_step"], res["fp16_runs"] = _median_replay_ms(m, side)
    torch.cuda.current_stream().wait_stream(side)
    res["fp16_note"] = "float16 operands (IEEE half MFMA), fp32 accumulation and residual stream; the folded pre-norm (bf16 build only) is off in this build"
    del m
    torch.cuda.empty_cache()
    # --- one external anchor (context, not a target): upstream's LTX-2 19 B audio+video model is quoted at 1.22 s/step on H100 for 121 frames
    #     at 720p, Euler, CFG = 1 (docs/LTX_2_Technical_Report_compressed.pdf section 6.3).  The 19B-style AudioVideo engine at the nearest valid latent grid:
    #     121 frames -> 16 latent frames, 1280 x 704 px -> 22 x 40 latent positions (720 is not a multiple of 32), 121 audio latents (4.84 s at 25 per second)
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_layers=layers, device=dev)
    m.init_random_weights(seed=0)
    gh = torch.Generator(device=dev).manual_seed(5)
    Nh, Nah, Sh = 16 * 22 * 40, 121, 1024
    vlh, alh = torch.randn(Nh, 128, generator=gh, device=dev), torch.randn(Nah, 128, generator=gh, device=dev)
    vch, ach = 0.1 * torch.randn(1, Sh, 3840, generator=gh, device=dev), 0.1 * torch.randn(1, Sh, 3840, generator=gh, device=dev)
    vph = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 16, 22, 40), fps=25.0).create_initial_state(device=dev).positions
    aph = AudioLatentTools(AudioPatchifier(1), AudioLatentShape(1, 8, Nah, 16)).create_initial_state(device=dev).positions
    m.prepare(vch, vph, audio_context=ach, audio_positions=aph)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(vlh, DISTILLED_SIGMA_VALUES, audio_latent=alh)
        ms, runs = _median_replay_ms(m, side)
    torch.cuda.current_stream().wait_stream(side)
    res["h100_table_geometry_av_ms_per_step"] = ms
    res["h100_table_geometry_av_runs"] = runs
    res["h100_table_geometry_av"] = {
        "geometry": "121 frames, 1280x704 px -> 16 x 22 x 40 = 14080 video tokens + 121 audio tokens, S = 1024, LTX-2 19B-style AudioVideo DiT, "
                    f"{layers} layers, bf16, Euler, CFG = 1 (one model evaluation per step), random-init weights",
        "video_stream_algorithmic_tflop": round(dit_algorithmic_flops(Nh, Sh, 4096, layers) / 1e12, 1),
        "published_h100_s_per_step": 1.22,
        "note": "context, not a target: upstream PyTorch on H100 (720p exactly, their kernels, their attention); here the same model family on one MI355X",
    }
    del m
    torch.cuda.empty_cache()
    # --- config 4 shape: 48-layer AudioVideo DiT with 9-row AdaLN, prompt-modulated text K/V, per-head gates
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_layers=layers, caption_channels=None, cross_attention_adaln=True,
                 apply_gated_attention=True, av_ca_timestep_scale_multiplier=1000, device=dev)
    m.init_random_weights(seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    N, Na, S = 3456, 68, 1024
    vlat, alat = torch.randn(N, 128, generator=g, device=dev), torch.randn(Na, 128, generator=g, device=dev)
    vctx, actx = 0.1 * torch.randn(1, S, 4096, generator=g, device=dev), 0.1 * torch.randn(1, S, 2048, generator=g, device=dev)
    vpos = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 9, 16, 24), fps=25.0).create_initial_state(device=dev).positions
    apos = AudioLatentTools(AudioPatchifier(1), AudioLatentShape(1, 8, Na, 16)).create_initial_state(device=dev).positions
    m.prepare(vctx, vpos, audio_context=actx, audio_positions=apos)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(vlat, DISTILLED_SIGMA_VALUES, audio_latent=alat)
        res["ltx23_audiovideo_ms_per_step"], res["ltx23_audiovideo_runs"] = _median_replay_ms(m, side)
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
    med, runs, lat = _median_s(lambda: pipe(ctx, None, conf))
    res["two_stage_1536x1024x65_denoise_upscale_s"] = round(med, 3)
    res["two_stage_1536x1024x65_denoise_upscale_runs"] = [round(r, 3) for r in runs]
    med, runs, fr = _median_s(lambda: decode_latent(lat, dec))
    res["two_stage_1536x1024x65_decode_frames_per_sec"] = round(fr.shape[0] / med, 1)
    res["two_stage_1536x1024x65_decode_runs_s"] = [round(r, 3) for r in runs]
    # the decode the pipeline ITSELF takes at this size (9 x 32 x 48 = 13 824 latent voxels > 4000: DistilledConfig._get_tiling_config
    # -> decode_tiled with the reference's default tiling, pipelines/distilled.py:217-219)
    tc = conf._get_tiling_config()
    assert tc is not None
    med, runs, vid = _median_s(lambda: next(decode_tiled(lat, dec, tc)))
    res["two_stage_1536x1024x65_decode_tiled_frames_per_sec"] = round(vid.shape[2] / med, 1)
    res["two_stage_1536x1024x65_decode_tiled_runs_s"] = [round(r, 3) for r in runs]
    res["two_stage_decode_note"] = ("`decode_tiled` is what DistilledPipeline runs at this size (reference default tiling: overlapping tiles decode "
                                    "~3.6x the volume); the whole-volume `decode_latent` figure is the same decoder without tiling")
    # --- one whole generation at the headline size with everything resident (the reference's docs/USAGE.md:310-314 quotes ~2 min end to end
    #     for 512 x 768 x 65, 8 steps, on an M3 Max): prompt setup + the 8-step loop + VAE decode to uint8 frames through DistilledPipeline; the text encoder
    #     (Gemma, not built) and the mp4 writer are outside it
    shp1 = VideoLatentShape(1, 128, 9, 16, 24)
    pat1 = VideoLatentPatchifier(1)
    pos1 = VideoLatentTools(pat1, shp1, fps=24.0).create_initial_state(device=dev).positions
    side1 = torch.cuda.Stream()

    def one_generation():
        # what scripts/generate.py's default path does after the text encoder: noise, prompt setup, the captured 8-step loop, unpatchify, decode
        c1 = ctx.clone()                                   # (a new tensor: the per-prompt caches are rebuilt, as for a new prompt)
        z = torch.randn(3456, 128, generator=g, device=dev)
        m.prepare(c1, pos1)
        side1.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side1):
            m.capture_denoise_graph(z, DISTILLED_SIGMA_VALUES)
            m.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side1)
        return decode_latent(pat1.unpatchify(z[None], shp1).contiguous(), dec)
    med, runs, fr = _median_s(one_generation)
    res["e2e_generate_s"] = round(med, 3)
    res["e2e_generate_runs_s"] = [round(r, 3) for r in runs]
    res["e2e_generate_note"] = (f"768x512x65: prompt setup + hipGraph capture + 8 distilled steps + VAE decode -> {tuple(fr.shape)} uint8 frames, weights resident, text features given; "
                                "the reference quotes ~2 min on an M3 Max for the same size (incl. its text encoder)")
    del m, dec, up, pipe
    torch.cuda.empty_cache()
    return res


def socket_power_during(work, seconds: float = 1.5, smi_device: int = 0):
    """Poll `rocm-smi --showpower` from a thread while `work()` is called in a loop for `seconds`: {mean_w, max_w, cap_w, samples}, or
    None when rocm-smi is unavailable.  Never inside a timed region."""
    import re
    import threading
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "-d", str(smi_device), "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            except Exception:  # noqa: BLE001
                return
            for line in out.splitlines():
                if line.startswith(f"card{smi_device}"):
                    try:
                        samples.append(float(line.split(",")[-1]))
                    except ValueError:
                        pass
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        work()
    stop[0] = True
    th.join(timeout=6)
    if len(samples) < 3:
        return None
    busy = samples[len(samples) // 3:]        # the reading ramps for about a second after the load starts
    cap = None
    try:
        out = subprocess.run(["rocm-smi", "-d", str(smi_device), "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", out)
        cap = float(m.group(1)) if m else None
    except Exception:  # noqa: BLE001
        pass
    return {"mean_w": round(sum(busy) / len(busy), 1), "max_w": max(busy), "cap_w": cap, "samples": len(busy), "smi_device": smi_device,
            "how": "rocm-smi --showpower polled during an extra hipGraph replay of the denoise loop, outside the timed regions"}


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` with no launcher environment: become the launcher (one rank per GPU, 127.0.0.1 rendezvous)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="the driver's value")
    ap.add_argument("--warmup", type=int, default=5, help="the driver's value")
    ap.add_argument("--layers", type=int, default=48, help="debug only; the headline config is 48")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="no hipGraph anywhere: K eager steps are timed (rocprofv3 --pmc passes, per-dispatch traces)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (AudioVideo step, two-stage pipeline)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi socket-power samples taken during an extra graph replay")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the second (instrumented) pass that times the dominant GEMM")
    ap.add_argument("--loader-layers", type=int, default=8, help="layers of the synthetic checkpoint the loader-throughput leg writes and loads "
                    "(8 = 4.3 GB, bounded for the default run; 48 = the full 25.8 GB file)")
    ap.add_argument("--no-loader", action="store_true", help="skip the checkpoint-loader throughput leg")
    ap.add_argument("--fold", type=int, default=None, help="engine option fold_norms for the headline model (0 = round 5's norm passes; default: the engine's, 1)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))

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
    if args.fold is not None:
        try:
            model.set_option("fold_norms", args.fold)
        except Exception as e:  # noqa: BLE001  (an A/B library of an earlier round has no such option)
            print(f"bench: fold_norms not set: {e}", file=sys.stderr)
    model.init_random_weights(seed=0, fill=(rank == 0))         # ranks > 0 only ALLOCATE (whatever the allocator hands back): their weights arrive by the broadcast
    wt = model.weight_tensors()
    w_bytes = sum(t.numel() * t.element_size() for t in wt.values())
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.time()
    bcast_mode = os.environ.get("LTX2_BCAST", "ring")
    n_coll = D.broadcast_tensors(wt, src=0, bucket_bytes=1 << 30, mode=bcast_mode)
    torch.cuda.synchronize()
    D.barrier()
    bcast_s = time.time() - t0
    # every replica holds the same bytes: MIN and MAX over ranks of a 63-bit checksum of all weight tensors agree
    weights_identical = D.replicas_identical(wt, dev) if world > 1 else None

    # ---------------- per-rank prompt / seed ----------------
    shape = VideoLatentShape.from_pixel_shape(VideoPixelShape(1, 65, 512, 768, 24.0))
    N = shape.frames * shape.height * shape.width
    S, Dm = 1024, model.inner_dim
    gen_lat = torch.Generator(device=dev).manual_seed(1234 + rank)
    tools = VideoLatentTools(VideoLatentPatchifier(1), shape, fps=24.0)
    state = tools.create_initial_state(device=dev)
    noise = torch.randn(N, 128, generator=gen_lat, device=dev)
    ctx = 0.1 * torch.randn(1, S, 3840, generator=gen_lat, device=dev)
    t0 = time.time()
    model.prepare(ctx, state.positions)
    torch.cuda.synchronize()
    prep_ms = (time.time() - t0) * 1e3

    sig = DISTILLED_SIGMA_VALUES
    K, W = args.steps, args.warmup
    lat = noise.clone()
    ts_dev = torch.tensor(sig[:8], device=dev)

    def run_steps(n):
        for i in range(n):
            s0, s1 = sig[i % 8], sig[i % 8 + 1]
            if i % 8 == 0:
                lat.copy_(noise)
            m = Modality(latent=lat[None], context=ctx, context_mask=None, timesteps=ts_dev[i % 8:i % 8 + 1], positions=state.positions)
            model.denoise_step_(lat, m, s0, s1)

    # The hot path has two launch forms of the SAME kernels: K calls of the fused step (ltx2_dit_denoise_step: what the pipelines run with a per-step
    # callback or guidance) and the replay of the captured 8-step hipGraph (pipelines/common.py use_hip_graph=True: one call per 8 steps).
    # `value` is ALWAYS the first form -- it exists for every K (the driver's K = 20 is not a multiple of 8), so rounds stay comparable;
    # the graph form is timed beside it over the nearest multiple of 8 steps and reported as hipgraph_ms_per_step.  Measured in round 4: with the
    # warm-up steps directly in front of the timed regions the two forms are within 0.4 % of each other (76.24 / 76.55, 76.30 / 76.56 ms per step).
    side = torch.cuda.Stream()

    def run_graph(n_steps):
        for _ in range(n_steps // 8):
            model.replay_denoise_graph()

    def timed(fn, n):
        """exactly n steps between barrier + synchronize on both sides -> (this rank's seconds, max over ranks)"""
        D.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t_start            # this rank's own steps (before the closing barrier)
        D.barrier()
        return mine, D.max_over_ranks(time.perf_counter() - t_start, dev)

    # set-up first (one eager step so every lazily sized buffer exists, then the capture), the W warm-up steps LAST: the timed region starts on a socket
    # that has just been running the same kernels (a timed region that follows the capture's idle time directly reads ~1 % slow)
    run_steps(1)
    torch.cuda.synchronize()
    graph_err = None
    if not args.no_graph:
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                lat.copy_(noise)
                model.capture_denoise_graph(lat, sig)
                model.replay_denoise_graph()            # warm-up replay
            side.synchronize()
        except Exception as e:  # noqa: BLE001
            graph_err = f"failed: {e}"
    else:
        graph_err = "skipped (--no-graph)"
    run_steps(W)
    torch.cuda.synchronize()

    # ---------------- THE timed region: exactly K steps, nothing else on the stream ----------------
    dt_rank, dt = timed(run_steps, K)
    n_joined = D.count_ranks(dev)                       # counted through the process group (an RCCL all-reduce on device tensors)

    # ---------------- the headline line exists from here on; every later leg can only ADD to it ----------------
    steps_per_s = world * K / dt
    ms_per_step = dt / K * 1e3
    alg = dit_algorithmic_flops(N, S, Dm, L)
    exe = dit_executed_flops(N, S, Dm, L)
    out = {
        "metric": "denoise_steps_per_sec", "value": round(steps_per_s, 4), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"LTX-2 19B distilled DiT ({L} layers, D=4096, 32x128 heads), 768x512x65 "
                               f"(N={N} video tokens, S={S} text tokens), 8-step distilled sigmas, bf16, random-init weights",
                   "parallelism": f"prompt-parallel x{world} (independent prompt/seed per GPU, one RCCL weight broadcast)"},
        "timed_with": "K ltx2_dit_denoise_step calls on the in-order stream (the headline form every round; the hipGraph replay of the same "
                      "kernels is hipgraph_ms_per_step)",
        "eager_ms_per_step": round(ms_per_step, 3),
        "step_algorithmic_tflop": round(alg / 1e12, 3),
        "step_mfma_roofline_frac": round(alg / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "step_executed_tflop": round(exe / 1e12, 3),
        "step_executed_mfma_frac": round(exe / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "step_executed_note": "executed = algorithmic minus the text cross-attention K / V projections (4 S D^2 per layer), which are step-invariant "
                              "and run once per prompt (prompt_setup_ms); the reference recomputes them every step",
        "prompt_setup_ms": round(prep_ms, 1),
        "rccl_ranks": n_joined, "rccl_ranks_how": "all_reduce(sum) of a device-resident 1 over the process group",
        "collective_backend": (torch.distributed.get_backend() if world > 1 else None), "weight_bytes": w_bytes, "weight_broadcast_s": round(bcast_s, 3),
        "weight_broadcast_collectives": n_coll, "weight_broadcast_mode": bcast_mode if world > 1 else None,
        "weights_identical": weights_identical,
        "weight_broadcast_gbps": round(w_bytes / bcast_s / 1e9, 1) if world > 1 and bcast_s > 0 else None,
    }
    emitted = [False]

    def emit():
        if rank == 0 and not emitted[0]:
            emitted[0] = True
            _REAL_STDOUT.write(json.dumps(out) + "\n")
            _REAL_STDOUT.flush()

    def leg(name, fn):
        """One secondary leg: whatever it raises becomes `<name>_error` on the line (a failure after the timed region must never
        cost the headline).  Legs hold NO collectives -- ranks meet only at the fixed points below -- so one rank's failure cannot hang the others."""
        try:
            if os.environ.get("LTX2_BENCH_FAIL_LEG") == name:       # tests/test_cli_gpu.py::test_bench_line_survives_a_failing_leg
                raise RuntimeError(f"LTX2_BENCH_FAIL_LEG={name}")
            return fn()
        except Exception as e:  # noqa: BLE001
            out[f"{name}_error"] = f"{type(e).__name__}: {e}"
            sys.stderr.write(f"[bench] leg {name} failed on rank {rank}: {type(e).__name__}: {e}\n")
            return None

    try:
        # ---------------- second pass (not part of `value`): HIP events around every launch of the dominant GEMM ----------------
        DOM_EPI = nv.EPI_RESID_GATE_F32     # gemm_v4_kernel<EPI_RESID_GATE_F32, 3, 224>: attn1.to_out, attn2.to_out, ff.net.2

        def kernel_pass():
            model.profile_begin(DOM_EPI)
            run_steps(K)
            torch.cuda.synchronize()
            return model.profile_end()
        k_ms, k_n, k_fl = (0.0, 0, 0.0) if args.no_kernel_pass else (leg("kernel_pass", kernel_pass) or (0.0, 0, 0.0))
        # ---------------- same-box A/B of round 6's folded norms (engine option fold_norms; not part of `value`): the same K eager steps with a norm pass in
        #                  front of every projection (round 5's form), on this box, right behind the headline ----------------
        def fold_ab():
            lvl = 1 if args.fold is None else args.fold
            try:
                model.set_option("fold_norms", 0)
                run_steps(max(W, 2))
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                run_steps(K)
                torch.cuda.synchronize()
                return (time.perf_counter() - t_start) / K * 1e3
            finally:
                model.set_option("fold_norms", lvl)
                run_steps(1)
        f0 = None if args.no_kernel_pass else leg("fold_ab", fold_ab)
        out["fold_norms"] = {"level": 1 if args.fold is None else args.fold, "unfolded_ms_per_step": None if f0 is None else round(f0, 3),
                             "note": "level 1 (default): the text cross-attention's RMS pre-norm rides on attn1.to_out's epilogue and attn2.to_q's accumulators (DESIGN.md); "
                                     "unfolded = option 0, the same K eager steps on this box"}
        # HBM/fabric traffic of the dominant kernel cannot be collected from inside the process: it comes from the committed
        # rocprofv3 --pmc passes of THIS kernel version (the newest profiles/r*_pmc_traffic.json names the commit), null if absent.
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            import glob
            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]) as f:
                tj = json.load(f)
                traffic, traffic_src = tj.get("per_launch_avg_bytes"), tj.get("commit")
                # stale = the kernel's sources changed since the counters were collected (files without the digest predate round 4: stale)
                traffic_stale = tj.get("kernel_source_sha16") != kernel_source_sha()
        except Exception:  # noqa: BLE001
            pass
        kern_avg_ms = k_ms / max(k_n, 1)
        kern_tflops = (k_fl / max(k_n, 1)) / (kern_avg_ms * 1e-3) / 1e12 if k_n else None
        out["roofline"] = {
            "kernel": "gemm_v4_kernel<EPI_RESID_GATE_F32, layout 3, 224> (224x256x64 tile, 4 waves, generated asm K loop, "
                      "v_mfma_f32_16x16x32_bf16: attn1/attn2 to_out and ff.net.2 + bias + gate * (.) added into the fp32 residual; two instantiations since round 6: "
                      "<4, 3, 224, false, 0> and, for attn1.to_out with the folded pre-norm's bf16 shadow + sums of squares, <4, 3, 224, false, 30>)",
            "bound": "mfma", "achieved": None if kern_tflops is None else round(kern_tflops, 1), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": None if kern_tflops is None else round(kern_tflops / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_source_commit": traffic_src, "traffic_stale": traffic_stale,
            "launches": k_n, "avg_launch_us": round(kern_avg_ms * 1e3, 2),
            "algorithmic_flops_per_launch": round(k_fl / max(k_n, 1)),
            "measured": "HIP events around every launch of this kernel in a separate pass of the same K steps (not in `value`)"}

        # ---------------- the hipGraph form over the nearest multiple of 8 steps (this rank's own clock), then socket power under it ----------------
        def graph_leg():
            if graph_err:
                raise RuntimeError(graph_err)
            n_g = max(8, K // 8 * 8)
            with torch.cuda.stream(side):
                run_graph(8)
                side.synchronize()
                t_start = time.perf_counter()
                run_graph(n_g)
                side.synchronize()
                ms = (time.perf_counter() - t_start) / n_g * 1e3
            torch.cuda.current_stream().wait_stream(side)
            return ms, n_g
        graph_ms, graph_steps = leg("hipgraph", graph_leg) or (None, None)

        def power_leg():
            # socket power while the same graph keeps replaying (~3 s, outside every timed region): the step time on this part is set by the
            # 1400 W cap, so the line carries the evidence; every rank samples ITS socket while all ranks keep replaying
            if graph_err:
                raise RuntimeError(graph_err)
            with torch.cuda.stream(side):
                pw = socket_power_during(lambda: (model.replay_denoise_graph(), side.synchronize()), seconds=3.0, smi_device=local)
            torch.cuda.current_stream().wait_stream(side)
            return pw
        power = None if args.no_power else leg("socket_power", power_leg)
        out["hipgraph_ms_per_step"] = None if graph_ms is None else round(graph_ms, 3)
        out["hipgraph_steps_timed"] = graph_steps
        out["socket_power"] = power

        # ---------------- VAE decode: latent in HBM -> uint8 frames in HBM ----------------
        def vae_leg():
            dec = SimpleVideoDecoder(device=dev)
            dec.init_random_weights(seed=7)
            dec.generator = torch.Generator(device=dev).manual_seed(99 + rank)
            gen_vae = torch.Generator(device=dev).manual_seed(4321 + rank)
            z = torch.randn(1, 128, shape.frames, shape.height, shape.width, generator=gen_vae, device=dev)
            frames = decode_latent(z, dec)                    # warm-up (also sizes the workspace)
            torch.cuda.synchronize()
            reps = 3
            t_start = time.perf_counter()
            for _ in range(reps):
                frames = decode_latent(z, dec)
            torch.cuda.synchronize()
            if tuple(frames.shape) != (65, 512, 768, 3):
                raise RuntimeError(f"decode_latent returned {tuple(frames.shape)}")
            return (time.perf_counter() - t_start) / reps
        D.barrier()
        vae_s = None if args.no_vae else leg("vae", vae_leg)

        # ---------------- fixed meeting point of the ranks: per-rank figures of every leg ----------------
        per_rank = D.gather_floats([dt_rank / K * 1e3, (power or {}).get("mean_w", -1.0), (power or {}).get("max_w", -1.0),
                                    -1.0 if graph_ms is None else graph_ms, -1.0 if vae_s is None else vae_s], dev)
        if rank != 0:
            return
        out["per_rank_ms_per_step"] = {"min": round(min(r[0] for r in per_rank), 3), "max": round(max(r[0] for r in per_rank), 3),
                                       "all": [round(r[0], 3) for r in per_rank]}
        out["per_rank_socket_power_w"] = None if args.no_power else {"mean": [r[1] for r in per_rank], "max": [r[2] for r in per_rank]}
        if world > 1:
            out["per_rank_hipgraph_ms_per_step"] = [None if r[3] < 0 else round(r[3], 3) for r in per_rank]
        vae_all = [r[4] for r in per_rank]
        if not args.no_vae and all(v > 0 for v in vae_all):
            vdt = max(vae_all)                                  # the slowest rank's decode: whole-job frames/s = world * 65 / that
            out["vae_decode_frames_per_sec"] = round(world * 65 / vdt, 2)
            out["vae_decode_ms"] = round(vdt * 1e3, 2)
            out["vae_roofline"] = {
                "workload": "decode_latent 768x512x65 (7/2 temporal chunking, cross-fade, uint8), latent and frames in HBM",
                "algorithmic_tflop": VAE_ALG_TFLOP, "algorithmic_gb": VAE_ALG_GB,
                "mfma": {"achieved": round(VAE_ALG_TFLOP / vdt, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(VAE_ALG_TFLOP / vdt / PEAK_BF16_TFLOPS, 4)},
                "hbm": {"achieved": round(VAE_ALG_GB / vdt, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(VAE_ALG_GB / vdt / PEAK_HBM_GBS, 4)},
                "bound": "mfma", "note": "the conv stack is MFMA-bound (arithmetic intensity ~2000 F/B); the HBM fraction cannot exceed ~0.15 (SURVEY 8d)"}
        else:
            out["vae_decode_frames_per_sec"], out["vae_decode_ms"] = None, None
        if world == 1 and not args.no_extra:
            # Secondary BASELINE configurations, reported beside the headline (never part of `value`): config 3 (fp8), config 4 shape
            # (LTX-2.3-style AudioVideo DiT, joint audio+video step) and config 5 (two-stage 1536x1024x65 with the spatial upscaler).
            def extra_leg():
                nonlocal model
                model = None                                    # the 38 GB of headline weights (every later leg builds its own model)
                torch.cuda.empty_cache()
                return extra_configs(dev, L)
            out["extra_configs"] = leg("extra_configs", extra_leg)
        if world == 1 and not args.no_loader:
            def loader_leg():
                torch.cuda.empty_cache()
                return loader_throughput(dev, args.loader_layers)
            out["loader"] = leg("loader", loader_leg)
        if world == 1 and not args.no_cpu_baseline:        # the CPU leg is reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = leg("cpu_baseline", lambda: cpu_baseline(min(CPU_THREADS, os.cpu_count() or 1)))
    finally:
        emit()


_REAL_STDOUT = sys.stdout

if __name__ == "__main__":
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # library progress lines go to stderr: stdout carries exactly ONE JSON line
        main()
