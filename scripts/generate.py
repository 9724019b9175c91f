#!/usr/bin/env python
"""LTX-2 generation CLI on MI355X for the distilled denoise + VAE-decode hot path.

Keeps the flag names and `generate_video(...)` keyword names of the reference's
scripts/generate.py (argparse block :2364-2641, `generate_video` :933-997) for this path:
standard single-stage distilled loop (reference :1764-1984) followed by `decode_latent` (:2080-2091).
Out of this path (and rejected with a clear message): Gemma text encoding, audio VAE / vocoder / muxing, CFG/STG guidance.
`--pipeline distilled --spatial-upscaler-weights W` runs the two-stage DistilledPipeline; adding `--generate-audio` runs it on
the AudioVideo transformer and saves the audio LATENT beside the frames.
`save_video` keeps the reference's ffmpeg settings (frames piped as raw RGB; PNG frames when no ffmpeg binary exists).  `--lora` fuses an adapter into the checkpoint weights at load.  `--image` conditions latent frame 0 on an image through the VAE encoder (the reference
routes that through its pipelines, scripts/generate.py:1711-1731).  Text embeddings come from `--embedding file.npz` (keys
`embedding`, `attention_mask`, as the reference's `load_text_embedding` :730-750) or the reference's
dummy encoder (`--no-gemma`, :642-661).  Frames are written as `<output>.npz` (uint8 T,H,W,3) and
the final latent as `<output>_latent.npz` like the reference (:1994-1996).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, VideoLatentPatchifier, get_pixel_coords, get_sigma_schedule  # noqa: E402
from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model  # noqa: E402
from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, TilingConfig, decode_latent, decode_tiled, load_vae_decoder_weights  # noqa: E402
from ltx_2_mlx_amd.model.video_vae_encoder import SimpleVideoEncoder, load_vae_encoder_weights  # noqa: E402
from ltx_2_mlx_amd.types import SpatioTemporalScaleFactors, VideoLatentShape  # noqa: E402


def create_dummy_text_encoding(prompt: str, batch_size: int = 1, max_tokens: int = 256, embed_dim: int = 3840, device="cuda"):
    """0.1 * N(0,1) embedding seeded by the prompt (reference :642-661; torch RNG instead of MLX's)."""
    g = torch.Generator().manual_seed(hash(prompt) % (2 ** 31))
    return (0.1 * torch.randn(batch_size, max_tokens, embed_dim, generator=g)).to(device), torch.ones(batch_size, max_tokens, device=device)


NATIVE_FPS = 24     # the model generates motion at 24 fps


def video_filters(fps: int = 24, speed: float = 1.0):
    """ffmpeg -vf chain of the reference's save_video (scripts/generate.py:2178-2194): speed first (setpts), then
    motion-compensated interpolation when the target frame rate exceeds the native 24 fps."""
    filters = []
    if speed != 1.0:
        filters.append(f"setpts={1.0 / speed}*PTS")
    if fps > NATIVE_FPS:
        filters.append(f"minterpolate=fps={fps}:mi_mode=mci:mc_mode=aobmc:me_mode=bidir:vsbmc=1")
    return filters


def ffmpeg_command(width: int, height: int, output_path: str, fps: int = 24, speed: float = 1.0):
    """Same encoder settings as the reference (:2201-2222: libx264, yuv420p, crf 18, input at the native 24 fps); the
    frames arrive as raw RGB on stdin instead of a directory of PNGs."""
    cmd = ["ffmpeg", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{width}x{height}", "-framerate", str(NATIVE_FPS), "-i", "-"]
    filters = video_filters(fps, speed)
    if filters:
        cmd += ["-vf", ",".join(filters)]
    return cmd + ["-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", "18", "-loglevel", "error", output_path]


def save_video(frames, output_path: str, fps: int = 24, speed: float = 1.0):
    """frames: uint8 (T, H, W, 3) array or list of (H, W, 3) arrays -> output_path via ffmpeg (reference :2153-2226).
    Without an ffmpeg binary the frames are written as PNGs into `<output stem>_frames/` and that directory is returned
    (the reference raises there; on a GPU box without ffmpeg the frames are still wanted)."""
    import shutil
    import subprocess
    frames = np.ascontiguousarray(np.stack([np.asarray(f) for f in frames]) if isinstance(frames, (list, tuple)) else np.asarray(frames))
    if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
        raise ValueError(f"save_video expects uint8 frames (T, H, W, 3), got {frames.dtype} {frames.shape}")
    t, h, w, _ = frames.shape
    if shutil.which("ffmpeg") is None:
        from PIL import Image
        out_dir = os.path.splitext(output_path)[0] + "_frames"
        os.makedirs(out_dir, exist_ok=True)
        for i in range(t):
            Image.fromarray(frames[i]).save(os.path.join(out_dir, f"frame_{i:04d}.png"))
        print(f"  ffmpeg not found: wrote {t} PNG frames to {out_dir}/")
        return out_dir
    result = subprocess.run(ffmpeg_command(w, h, output_path, fps, speed), input=frames.tobytes(), capture_output=True)
    if result.returncode != 0:
        raise RuntimeError(f"FFmpeg failed: {result.stderr.decode(errors='replace')}")
    return output_path


def load_text_embedding(path: str, device="cuda"):
    z = np.load(path)
    emb = torch.from_numpy(z["embedding"]).float()
    if emb.dim() == 2:
        emb = emb[None]
    mask = torch.from_numpy(z["attention_mask"]).float() if "attention_mask" in z else torch.ones(emb.shape[:2])
    return emb.to(device), mask.to(device)


def encode_text_features(path: str, weights_path=None, device="cuda", seed: int = 0):
    """`--text-features file.npz`: key `features` = the Gemma feature extractor's output [T, 3840] or [1, T, 3840]
    (or `hidden_states` [L, T, 3840] = all Gemma layers, run through the extractor first) and `attention_mask` [T].
    The Embeddings1DConnector (registers, 2 blocks, final RMSNorm) runs on the GPU (reference `encode_projected` /
    `encode_from_hidden_states`, model/text_encoder/encoder.py:136-215); its weights come from `--weights`
    (`model.diffusion_model.video_embeddings_connector.*`, `text_embedding_projection.*`) or are random without one."""
    from ltx_2_mlx_amd.model.text_encoder import create_text_encoder, load_text_encoder_weights
    z = np.load(path)
    enc = create_text_encoder(device=device)
    if weights_path:
        load_text_encoder_weights(enc, weights_path)
    else:
        enc.embeddings_connector.init_random_weights(seed)
    mask = torch.from_numpy(z["attention_mask"]).float().reshape(1, -1) if "attention_mask" in z else None
    if "hidden_states" in z:
        hs = [torch.from_numpy(h).float()[None] for h in z["hidden_states"]]
        mask = torch.ones(1, hs[0].shape[1]) if mask is None else mask
        out = enc.encode_from_hidden_states(hs, mask.to(device))
    else:
        feats = torch.from_numpy(z["features"]).float()
        feats = feats[None] if feats.dim() == 2 else feats
        mask = torch.ones(feats.shape[:2]) if mask is None else mask
        out = enc.encode_projected(feats.to(device), mask.to(device))
    return out.video_encoding, out.attention_mask.float()


def load_transformer(weights_path, num_layers=48, num_heads=32, caption_channels=3840, seed=0, device="cuda", use_fp8=False,
                     lora_path=None, lora_strength=1.0):
    """LTXModel(VideoOnly, 32x128, 48 layers, caption 3840) (reference load_transformer :788-835)."""
    model = LTXModel(num_attention_heads=num_heads, attention_head_dim=128, num_layers=num_layers,
                     caption_channels=caption_channels, device=device)
    if weights_path:
        from ltx_2_mlx_amd.loader import is_fp8_checkpoint, load_transformer_weights
        from ltx_2_mlx_amd.loader import LoRAConfig
        load_transformer_weights(model, weights_path, strict=True, use_fp8=use_fp8 or is_fp8_checkpoint(weights_path),
                                 lora_configs=[LoRAConfig(lora_path, lora_strength)] if lora_path else None)
    elif lora_path:
        raise ValueError("--lora needs --weights (an adapter is fused into checkpoint weights)")
    else:
        model.init_random_weights(seed=seed)
    return model


def load_av_transformer(weights_path, num_layers=48, num_heads=32, caption_channels=3840, seed=0, device="cuda", use_fp8=False):
    """AudioVideo LTXModel (video 32x128 + audio 32x64 heads; reference load_av_transformer :838-902)."""
    from ltx_2_mlx_amd.model.transformer import LTXModelType
    model = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=num_heads, attention_head_dim=128, num_layers=num_layers,
                     caption_channels=caption_channels, audio_attention_heads=num_heads, device=device)
    if weights_path:
        from ltx_2_mlx_amd.loader import is_fp8_checkpoint, load_av_transformer_weights
        load_av_transformer_weights(model, weights_path, strict=True, use_fp8=use_fp8 or is_fp8_checkpoint(weights_path))
    else:
        model.init_random_weights(seed=seed)
    return model


def euler_step_x0(sample, denoised, sigma, sigma_next):
    from ltx_2_mlx_amd import kernels as K
    c = sample.shape[-1]
    return K.euler_step(sample.reshape(-1, c).float(), denoised.reshape(-1, c).float(), sigma, sigma_next).reshape(sample.shape)


# Keyword surface of the reference's generate_video (scripts/generate.py:933-997), names and defaults verbatim, so a caller of
# the reference can switch without touching its call site.  Options outside the MI355X hot path are accepted at their
# reference defaults and raise NotImplementedError only when set to something else.
_OUT_OF_PATH_DEFAULTS = dict(
    upscale_temporal=False, temporal_upscaler_weights=None, early_layers_only=False, enhance_prompt_flag=False,
    cross_attn_scale=1.0, distilled_lora=None, stg_scale=0.0, apg_scale=1.0, control_video=None, save_control=False,
    ge_gamma=0.0, keyframes=None, ic_lora_weights=None, negative_prompt=None)


def generate_video(
    prompt: str,
    height: int = 480,
    width: int = 704,
    num_frames: int = 97,
    num_steps: int = 7,
    cfg_scale: float = 5.0,
    guidance_rescale: float = 0.7,
    steps_stage1: int = 15,
    steps_stage2: int = 3,
    cfg_stage1=None,
    seed: int = 42,
    weights_path=None,
    output_path: str = "gens/output.mp4",
    use_placeholder: bool = False,
    skip_vae: bool = False,
    embedding_path=None,
    gemma_path: str = "weights/gemma-3-12b",
    use_gemma: bool = True,
    use_fp16: bool = True,
    use_fp8: bool = False,
    model_variant: str = "distilled",
    upscale_spatial: bool = False,
    spatial_upscaler_weights=None,
    upscale_temporal: bool = False,
    temporal_upscaler_weights=None,
    generate_audio: bool = False,
    low_memory: bool = False,
    fast_mode: bool = False,
    image_path=None,
    image_strength: float = 0.95,
    lora_path=None,
    lora_strength: float = 1.0,
    tiled_vae: bool = False,
    pipeline_type: str = "text-to-video",
    early_layers_only: bool = False,
    enhance_prompt_flag: bool = False,
    cross_attn_scale: float = 1.0,
    distilled_lora=None,
    distilled_lora_scale: float = 1.0,
    stg_scale: float = 0.0,
    stg_mode: str = "video",
    apg_scale: float = 1.0,
    apg_eta: float = 1.0,
    apg_norm_threshold: float = 0.0,
    apg_momentum: float = 0.0,
    control_video=None,
    control_type: str = "raw",
    canny_low: int = 100,
    canny_high: int = 200,
    control_strength: float = 0.95,
    save_control: bool = False,
    ge_gamma: float = 0.0,
    output_fps: int = 24,
    output_speed: float = 1.0,
    keyframes=None,
    ic_lora_weights=None,
    audio_cfg_scale=None,
    rescale_scale=None,
    negative_prompt=None,
    *,
    # MI355X additions (keyword-only; the reference has no counterpart)
    use_hip_graph: bool = True,
    num_layers: int = 48,
    num_heads: int = 32,
    vae_base_channels: int = 128,
    device: str = "cuda",
    text_features_path=None,
    save_mp4: bool = True,
):
    """Generate video from a text prompt: distilled denoise loop + VAE decode on MI355X behind the reference's signature."""
    given = dict(upscale_temporal=upscale_temporal, temporal_upscaler_weights=temporal_upscaler_weights, early_layers_only=early_layers_only,
                 enhance_prompt_flag=enhance_prompt_flag and use_gemma, cross_attn_scale=cross_attn_scale, distilled_lora=distilled_lora,
                 stg_scale=stg_scale, apg_scale=apg_scale, control_video=control_video, save_control=save_control, ge_gamma=ge_gamma,
                 keyframes=keyframes, ic_lora_weights=ic_lora_weights, negative_prompt=negative_prompt)
    for k, v in given.items():
        if v != _OUT_OF_PATH_DEFAULTS[k]:
            raise NotImplementedError(f"{k}={v!r} is outside the MI355X hot path (see DESIGN.md); leave it at its default {_OUT_OF_PATH_DEFAULTS[k]!r}")
    if pipeline_type not in ("text-to-video", "distilled", "one-stage", "two-stage"):
        raise NotImplementedError(f"pipeline_type={pipeline_type!r} is outside the MI355X hot path (see DESIGN.md)")
    output_dir = os.path.dirname(output_path)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)          # reference :1000-1003
    if num_frames % 8 != 1:
        raise ValueError(f"num_frames must be 8*k + 1, got {num_frames}")
    if height % 32 != 0 or width % 32 != 0:
        raise ValueError(f"Resolution ({height}x{width}) must be divisible by 32")
    if not use_fp16:
        raise NotImplementedError("use_fp16=False (fp32 compute): the MI355X path computes in bf16 with fp32 accumulation and an fp32 residual stream")
    print("Compute dtype: bf16 operands / fp32 accumulate / fp32 residual stream (use_fp16=True selects the reduced-precision path; "
          "fp16 operands are not built)", file=sys.stderr)
    if use_gemma and not (embedding_path or text_features_path):
        if not os.path.exists(gemma_path):              # the reference prints this and returns (:1085-1092)
            print(f"\n  ERROR: Gemma weights not found at {gemma_path}\n  Use use_gemma=False (--no-gemma) for dummy embeddings, or pass "
                  f"embedding_path / text_features_path")
            return None
        raise NotImplementedError("Gemma-3 text encoding is outside the hot path: pass embedding_path / text_features_path (or use_gemma=False)")
    if model_variant == "distilled" and cfg_scale > 1.2 and pipeline_type != "two-stage":
        print(f"  WARNING: Distilled model requires CFG=1.0 (no guidance). You requested {cfg_scale}.\n  Forcing CFG=1.0 (reference :1207-1216).")
        cfg_scale, guidance_rescale = 1.0, 0.0
    if cfg_scale > 1.0:
        raise NotImplementedError(f"cfg_scale={cfg_scale}: classifier-free guidance is outside the MI355X hot path (distilled single-pass only)")
    if low_memory or fast_mode:
        print("  low_memory / fast_mode: no effect here (weights and caches stay resident in HBM, the loop is one hipGraph)")
    num_inference_steps = num_steps
    pipeline = "distilled" if (pipeline_type in ("distilled", "two-stage") or upscale_spatial) else "text-to-video"
    fps, speed = output_fps, output_speed
    torch.manual_seed(seed)
    t_all = time.time()
    print("[1/5] text encoding")
    if text_features_path:
        text_encoding, _ = encode_text_features(text_features_path, weights_path, device, seed)
    else:
        text_encoding, _ = load_text_embedding(embedding_path, device) if embedding_path else create_dummy_text_encoding(prompt, device=device)
    print("[2/5] transformer")
    if generate_audio:
        if not spatial_upscaler_weights:
            raise ValueError("--generate-audio runs the joint audio+video DistilledPipeline: it needs --spatial-upscaler-weights")
        if lora_path:
            raise NotImplementedError("--lora with --generate-audio")
        model = X0Model(load_av_transformer(weights_path, num_layers, num_heads, text_encoding.shape[-1], seed, device, use_fp8=use_fp8))
    else:
        model = X0Model(load_transformer(weights_path, num_layers, num_heads, text_encoding.shape[-1], seed, device, use_fp8=use_fp8,
                                         lora_path=lora_path, lora_strength=lora_strength))
    print("[3/5] VAE decoder")
    vae_decoder = None
    if not skip_vae:
        vae_decoder = SimpleVideoDecoder(base_channels=vae_base_channels, device=device)
        if weights_path:
            load_vae_decoder_weights(vae_decoder, weights_path)
        else:
            vae_decoder.init_random_weights(seed=seed + 1)
    if spatial_upscaler_weights or pipeline == "distilled":
        # `--pipeline distilled` + `--spatial-upscaler-weights`: the reference's two-stage DistilledPipeline
        # (pipelines/distilled.py:274-505; scripts/generate.py:1622-1700): 8 steps at half resolution, x2 latent
        # upscale, 3 steps at full resolution.  "random" as the path builds a random-weight upscaler (no checkpoint here).
        if not spatial_upscaler_weights:
            raise ValueError("--pipeline distilled needs --spatial-upscaler-weights (two-stage pipeline)")
        if image_path or tiled_vae:
            raise NotImplementedError("the two-stage pipeline here takes neither --image nor --tiled-vae (single-stage text-to-video does)")
        if vae_decoder is None:
            raise ValueError("the two-stage pipeline needs the VAE weights (per-channel statistics): drop --skip-vae")
        from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler, load_spatial_upscaler_weights
        from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
        mid = 1024 if vae_base_channels == 128 else 64
        up = SpatialUpscaler(mid_channels=mid, device=device) if spatial_upscaler_weights != "random" else \
            SpatialUpscaler(mid_channels=mid, num_blocks_per_stage=4 if mid == 1024 else 1, device=device)
        if spatial_upscaler_weights == "random":
            up.init_random_weights(seed=seed + 2)
        else:
            load_spatial_upscaler_weights(up, spatial_upscaler_weights)
        pipe = DistilledPipeline(model, vae_decoder, vae_decoder, spatial_upscaler=up)
        conf = DistilledConfig(height=height, width=width, num_frames=num_frames, seed=seed, fps=24.0, use_hip_graph=use_hip_graph,
                               audio_enabled=generate_audio)
        print("[4/5] two-stage distilled pipeline (8 steps at half resolution, x2 upscale, 3 steps)" + (" with the audio branch" if generate_audio else ""))
        t0 = time.time()
        base = os.path.splitext(output_path)[0]
        if generate_audio:
            # the audio text context: `audio_embedding` of the --embedding file when present, else the video context
            actx = text_encoding
            if embedding_path and "audio_embedding" in np.load(embedding_path):
                actx = torch.from_numpy(np.load(embedding_path)["audio_embedding"]).float().to(device)
                actx = actx[None] if actx.dim() == 2 else actx
            frames, audio_latent = pipe(text_encoding, None, conf, audio_encoding=actx)
            np.savez(base + "_audio_latent.npz", latent=audio_latent.float().cpu().numpy())     # audio VAE / vocoder are outside this path
        else:
            frames = pipe(text_encoding, None, conf)
        torch.cuda.synchronize()
        print(f"  two-stage: {(time.time() - t0):.3f} s -> {tuple(frames.shape)}")
        frames_np = frames.cpu().numpy()
        np.savez_compressed(base + ".npz", frames=frames_np)
        if save_mp4:
            print(f"  video: {save_video(frames_np, output_path, fps=fps, speed=speed)}")
        print(f"Done in {time.time() - t_all:.1f} s: {base}.npz")
        return frames
    print("[4/5] latent noise")
    lf, lh, lw = (num_frames - 1) // 8 + 1, height // 32, width // 32
    g = torch.Generator(device=device).manual_seed(seed)
    latent = torch.randn(1, 128, lf, lh, lw, generator=g, device=device)
    sigmas = DISTILLED_SIGMA_VALUES[:num_inference_steps + 1] if model_variant == "distilled" else \
        [float(s) for s in get_sigma_schedule(num_inference_steps, distilled=False, latent=latent)]
    patchifier = VideoLatentPatchifier(patch_size=1)
    shape = VideoLatentShape(1, 128, lf, lh, lw)
    coords = patchifier.get_patch_grid_bounds(shape, device=device)
    positions = get_pixel_coords(coords, SpatioTemporalScaleFactors.default(), causal_fix=True).float()
    positions = torch.cat([positions[:, 0:1] / 24.0, positions[:, 1:]], dim=1)          # fps = 24 on the CLI path (:1823)
    print(f"[5/5] denoising ({len(sigmas) - 1} steps)")
    t0 = time.time()
    tok = patchifier.patchify(latent).contiguous()
    if use_placeholder:
        for i in range(len(sigmas) - 1):
            tok = tok + 0.1 * torch.randn_like(tok) * (sigmas[i + 1] - sigmas[i])
    elif image_path:
        # image-to-video: encoded image replaces latent frame 0, its tokens keep (1 - strength) of the noise level
        from ltx_2_mlx_amd.conditioning import VideoLatentTools
        from ltx_2_mlx_amd.components import GaussianNoiser
        from ltx_2_mlx_amd.pipelines import DistilledPipeline, ImageCondition, apply_conditionings, create_image_conditionings
        enc = SimpleVideoEncoder(device=device)
        if weights_path:
            load_vae_encoder_weights(enc, weights_path)
        else:
            enc.init_random_weights(seed=seed + 2)
        tools = VideoLatentTools(patchifier, shape, fps=24.0)
        st = tools.create_initial_state(device=device)
        st = apply_conditionings(st, create_image_conditionings([ImageCondition(image_path, 0, image_strength)], enc, height, width), tools)
        st = GaussianNoiser()(st, noise_scale=1.0, noise=tok)
        st, _ = DistilledPipeline(model, enc, None)._denoise_loop_av(st, None, sigmas, text_encoding, use_hip_graph=use_hip_graph)
        tok = st.latent
    elif use_hip_graph:
        vm = model.velocity_model
        vm.prepare(text_encoding, positions)
        lat2d = tok[0].float().contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            vm.capture_denoise_graph(lat2d, sigmas)
            vm.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side)
        tok = lat2d[None]
    else:
        for i in range(len(sigmas) - 1):
            m = Modality(latent=tok, context=text_encoding, context_mask=None,
                         timesteps=torch.tensor([sigmas[i]], device=device), positions=positions, enabled=True)
            tok = euler_step_x0(tok, model(m), sigmas[i], sigmas[i + 1])
    torch.cuda.synchronize()
    print(f"  denoise: {(time.time() - t0):.3f} s")
    latent = patchifier.unpatchify(tok, shape)
    base = os.path.splitext(output_path)[0]
    np.savez(base + "_latent.npz", latent=latent.float().cpu().numpy())
    frames = None
    if vae_decoder is not None:
        t0 = time.time()
        if tiled_vae:
            video = next(decode_tiled(latent, vae_decoder, TilingConfig.default()))
            from ltx_2_mlx_amd import kernels as K
            frames = K.video_to_uint8(video[0])
        else:
            frames = decode_latent(latent, vae_decoder)
        torch.cuda.synchronize()
        print(f"  decode: {(time.time() - t0):.3f} s -> {tuple(frames.shape)}")
        frames_np = frames.cpu().numpy()
        np.savez_compressed(base + ".npz", frames=frames_np)
        if save_mp4:
            print(f"  video: {save_video(frames_np, output_path, fps=fps, speed=speed)}")
    print(f"Done in {time.time() - t_all:.1f} s: {base}.npz")
    return frames


def main():
    p = argparse.ArgumentParser(description="LTX-2 video generation (MI355X hot path)")
    p.add_argument("prompt", type=str)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--width", type=int, default=704)
    p.add_argument("--frames", type=int, default=97)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--output", "-o", type=str, default="output.mp4")
    p.add_argument("--weights", type=str, default=None)
    p.add_argument("--embedding", type=str, default=None)
    p.add_argument("--text-features", type=str, default=None, help="npz with Gemma `features` [T,3840] (or `hidden_states` [L,T,3840]) + `attention_mask`: run the text connector on the GPU")
    p.add_argument("--no-gemma", action="store_true", help="dummy text embeddings (reference default without Gemma weights)")
    p.add_argument("--gemma-path", type=str, default=None)
    p.add_argument("--model-variant", choices=["distilled", "dev"], default="distilled")
    p.add_argument("--pipeline", type=str, default="text-to-video")
    p.add_argument("--cfg", type=float, default=1.0)
    p.add_argument("--fp16", action="store_true", help="reference default; here: bf16 operands, fp32 accumulate (a notice is printed)")
    p.add_argument("--fp32", action="store_true")
    p.add_argument("--fp8", action="store_true")
    p.add_argument("--skip-vae", action="store_true")
    p.add_argument("--placeholder", action="store_true")
    p.add_argument("--tiled-vae", action="store_true")
    p.add_argument("--low-memory", action="store_true")
    p.add_argument("--fast-mode", action="store_true")
    p.add_argument("--no-hip-graph", action="store_true", help="MI355X: run the step loop eagerly instead of replaying the captured hipGraph")
    p.add_argument("--image", type=str, default=None)
    p.add_argument("--image-strength", type=float, default=0.95)
    p.add_argument("--lora", type=str, default=None)
    p.add_argument("--lora-strength", type=float, default=1.0)
    p.add_argument("--generate-audio", action="store_true")
    p.add_argument("--spatial-upscaler-weights", type=str, default=None)
    p.add_argument("--layers", type=int, default=48, help="debug: number of DiT layers for random-weight runs")
    p.add_argument("--heads", type=int, default=32, help="debug: attention heads (x128) for random-weight runs")
    p.add_argument("--vae-base-channels", type=int, default=128)
    p.add_argument("--fps", type=int, default=24, help="output frame rate; > 24 interpolates (reference flag)")
    p.add_argument("--speed", type=float, default=1.0, help="playback speed multiplier (reference flag)")
    p.add_argument("--no-video-file", action="store_true", help="keep only the .npz outputs (skip ffmpeg / PNG frames)")
    a = p.parse_args()
    if a.pipeline not in ("text-to-video", "distilled", "one-stage", "two-stage"):
        raise NotImplementedError(f"--pipeline {a.pipeline} is outside the MI355X hot path")
    if a.model_variant == "dev" and a.cfg != 1.0:
        raise NotImplementedError("--model-variant dev with --cfg != 1: classifier-free guidance is not built on this path")
    generate_video(a.prompt, height=a.height, width=a.width, num_frames=a.frames, num_steps=a.steps, seed=a.seed, cfg_scale=a.cfg,
                   output_path=a.output, weights_path=a.weights, embedding_path=a.embedding, text_features_path=a.text_features,
                   gemma_path=a.gemma_path or "weights/gemma-3-12b", use_gemma=not a.no_gemma and not (a.embedding or a.text_features) and bool(a.gemma_path),
                   use_fp16=not a.fp32, model_variant=a.model_variant, skip_vae=a.skip_vae,
                   use_placeholder=a.placeholder, tiled_vae=a.tiled_vae, use_hip_graph=not a.no_hip_graph, use_fp8=a.fp8,
                   low_memory=a.low_memory, fast_mode=a.fast_mode, num_layers=a.layers, num_heads=a.heads, vae_base_channels=a.vae_base_channels,
                   image_path=a.image, image_strength=a.image_strength, lora_path=a.lora, lora_strength=a.lora_strength,
                   output_fps=a.fps, output_speed=a.speed, save_mp4=not a.no_video_file, generate_audio=a.generate_audio,
                   spatial_upscaler_weights=a.spatial_upscaler_weights, pipeline_type=a.pipeline)


if __name__ == "__main__":
    main()
