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


def _read_checkpoint_config(checkpoint_path: str) -> dict:
    """The JSON `config` entry of the safetensors metadata (reference scripts/generate.py:142-152); {} when absent / unreadable."""
    import json
    try:
        from safetensors import safe_open
        with safe_open(checkpoint_path, framework="pt") as f:
            metadata = f.metadata() or {}
        return json.loads(metadata.get("config", "{}"))
    except Exception:
        return {}


def detect_model_version(checkpoint_path: str) -> str:
    """`model_version` of the safetensors metadata, e.g. "2.3.0"; "" if unknown (reference :224-236)."""
    try:
        from safetensors import safe_open
        with safe_open(checkpoint_path, framework="pt") as f:
            metadata = f.metadata() or {}
        return metadata.get("model_version", "")
    except Exception:
        return ""


def is_v2_model(checkpoint_path: str) -> bool:
    """LTX-2.3 ("V2") checkpoint? (reference :239-242)"""
    return detect_model_version(checkpoint_path).startswith("2.3")


def get_vae_config(checkpoint_path: str) -> dict:
    """`config.vae` of the checkpoint metadata: decoder_blocks / decoder_base_channels / timestep_conditioning (reference :245-254)."""
    return _read_checkpoint_config(checkpoint_path).get("vae", {})


def create_vae_decoder(weights_path, device="cuda", seed=0, use_placeholder=False, base_channels_override=None, compute_dtype=None):
    """SimpleVideoDecoder built from the checkpoint's own architecture record (reference :1255-1273): decoder_blocks,
    decoder_base_channels (128 when absent), timestep_conditioning (True when absent).  Without a readable checkpoint the
    default 19B decoder is built and random-initialised (no checkpoints exist on a bare box; the reference would run the
    uninitialised module)."""
    have = bool(weights_path) and os.path.exists(weights_path)
    vae_config = get_vae_config(weights_path) if have else {}
    decoder_blocks = vae_config.get("decoder_blocks", None)
    base_channels = vae_config.get("decoder_base_channels", 128) if base_channels_override is None else base_channels_override
    timestep_cond = vae_config.get("timestep_conditioning", True)
    if decoder_blocks:
        print(f"  VAE config: {len(decoder_blocks)} blocks, base_ch={base_channels}, timestep={timestep_cond}")
    dec = SimpleVideoDecoder(decoder_blocks=decoder_blocks, base_channels=base_channels, timestep_conditioning=timestep_cond, device=device,
                             **({} if compute_dtype is None else {"compute_dtype": compute_dtype}))
    if have and not use_placeholder:
        load_vae_decoder_weights(dec, weights_path)
    else:
        if weights_path and not have:
            print(f"  Warning: Weights not found at {weights_path}, using random init")
        dec.init_random_weights(seed=seed)
    return dec


def _dtype_of(compute_dtype):
    return torch.bfloat16 if compute_dtype is None else compute_dtype


def load_transformer(weights_path, num_layers: int = 48, compute_dtype=None, use_fp8: bool = False, low_memory: bool = False,
                     fast_mode: bool = False, *, num_heads: int = 32, caption_channels: int = 3840, seed: int = 0, device="cuda",
                     lora_path=None, lora_strength: float = 1.0, fp8_resident: bool = False, fp8_compute: bool = False):
    """LTXModel(VideoOnly, 32x128, caption 3840) with checkpoint weights (reference load_transformer :788-835; same leading
    parameters).  A missing checkpoint file means random init with the reference's warning.  low_memory / fast_mode have
    no effect here; keyword-only extras are MI355X additions.  fp8_compute (BASELINE config 3, "fp8 weights (CDNA4 fp8 MFMA)"):
    the video stream's projections run fp8 x fp8 on the fp8 MFMA -- an fp8 checkpoint's codes stay resident (as fp8_resident),
    a bf16 / fp16 checkpoint is quantised per output channel at load; activations are quantised per token inside the step.  Not
    bit-identical to the reference's dequantise-at-load (`use_fp8`), which stays the default (reference :2424-2436,
    loader/fp8_loader.py:54-130)."""
    model = LTXModel(num_attention_heads=num_heads, attention_head_dim=128, num_layers=num_layers, caption_channels=caption_channels,
                     compute_dtype=_dtype_of(compute_dtype), device=device, fp8_compute=fp8_compute)
    if weights_path and os.path.exists(weights_path):
        from ltx_2_mlx_amd.loader import LoRAConfig, is_fp8_checkpoint, load_transformer_weights
        fp8 = use_fp8 or is_fp8_checkpoint(weights_path)
        # LoRA fusion needs dequantised weights; with fp8_compute they are re-quantised per channel after the fuse (load_state_dict)
        load_transformer_weights(model, weights_path, strict=True, use_fp8=fp8, fp8_resident=(fp8_resident or fp8_compute) and fp8 and not lora_path,
                                 lora_configs=[LoRAConfig(lora_path, lora_strength)] if lora_path else None)
    elif lora_path:
        raise ValueError("--lora needs --weights (an adapter is fused into checkpoint weights)")
    else:
        if weights_path:
            print(f"  Warning: Weights not found at {weights_path}, using random init")
        model.init_random_weights(seed=seed)
    return model


def load_av_transformer(weights_path, num_layers: int = 48, compute_dtype=None, use_fp8: bool = False, low_memory: bool = False,
                        caption_channels=3840, cross_attention_adaln: bool = False, apply_gated_attention: bool = False, *,
                        num_heads: int = 32, seed: int = 0, device="cuda", lora_path=None, lora_strength: float = 1.0,
                        fp8_compute: bool = False):
    """AudioVideo LTXModel (video 32x128 + audio 32x64 heads) with checkpoint weights (reference load_av_transformer :838-902,
    same parameters): caption_channels = 3840 for LTX-2.0, None for LTX-2.3 (the feature extractor already projects to the
    transformer widths); cross_attention_adaln / apply_gated_attention = the LTX-2.3 block variant;
    av_ca_timestep_scale_multiplier = 1000 as the reference passes it.  lora_path / lora_strength (keyword-only): the adapter is fused
    into whichever transformer is loaded, as the reference does (:1186-1202, loader/lora_loader.py:129-195); fp8_compute: as load_transformer
    (the VIDEO stream's projections; the audio stream and the cross-modal attention stay on the 16-bit path)."""
    from ltx_2_mlx_amd.model.transformer import LTXModelType
    model = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=num_heads, attention_head_dim=128, num_layers=num_layers,
                     caption_channels=caption_channels, audio_attention_heads=num_heads, cross_attention_adaln=cross_attention_adaln,
                     apply_gated_attention=apply_gated_attention, av_ca_timestep_scale_multiplier=1000,
                     compute_dtype=_dtype_of(compute_dtype), device=device, fp8_compute=fp8_compute)
    if weights_path and os.path.exists(weights_path):
        from ltx_2_mlx_amd.loader import LoRAConfig, is_fp8_checkpoint, load_av_transformer_weights
        fp8 = use_fp8 or is_fp8_checkpoint(weights_path)
        load_av_transformer_weights(model, weights_path, strict=True, use_fp8=fp8, fp8_resident=fp8_compute and fp8 and not lora_path,
                                    lora_configs=[LoRAConfig(lora_path, lora_strength)] if lora_path else None)
    elif lora_path:
        raise ValueError("--lora needs --weights (an adapter is fused into checkpoint weights)")
    else:
        if weights_path:
            print(f"  Warning: Weights not found at {weights_path}, using random init")
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
    upscale_temporal=False, early_layers_only=False, enhance_prompt_flag=False,
    cross_attn_scale=1.0, distilled_lora=None, stg_scale=0.0, apg_scale=1.0, control_video=None, save_control=False,
    ge_gamma=0.0, keyframes=None, ic_lora_weights=None, negative_prompt=None)
# pipelines whose algorithm is not built: "two-stage" is the dev model's CFG stage 1 + distilled-LoRA stage 2 (TwoStageCFGConfig,
# reference :1276-1431), "ic-lora" / "keyframe-interpolation" condition on control videos / keyframes (:1434-1636)
_PIPELINES_BUILT = ("text-to-video", "distilled", "one-stage")
_PIPELINES_KNOWN = _PIPELINES_BUILT + ("two-stage", "ic-lora", "keyframe-interpolation")


def _frames_from_video(video):
    """Pipeline output -> uint8 (T, H, W, 3): decode_latent already returns that; decode_tiled returns float (1, 3, T, H, W) in
    [-1, 1] (reference :1744-1755 converts it the same way)."""
    if video.dtype == torch.uint8:
        return video
    from ltx_2_mlx_amd import kernels as K
    return K.video_to_uint8(video[0] if video.dim() == 5 else video)


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
    vae_base_channels=None,
    device: str = "cuda",
    text_features_path=None,
    save_mp4: bool = True,
    two_stage_distilled: bool = False,
    fp8_resident: bool = False,
    model_version=None,
    compute_dtype=None,
    fp8_compute: bool = False,
):
    """Generate video from a text prompt: denoise loop + VAE decode on MI355X behind the reference's signature.

    Routing follows the reference (scripts/generate.py:1064-2095): an LTX-2.3 checkpoint (`model_version` 2.3.* in the
    safetensors metadata) or generate_audio=True runs the AudioVideo transformer through OneStagePipeline (fps 25,
    LTX2Scheduler over num_steps); everything else the standard video-only loop on the distilled sigma table; the VAE decoder is
    built from the checkpoint's `config.vae` record; upscale_spatial doubles the denoised latent before decoding (2W x 2H output).
    pipeline_type "two-stage" / "ic-lora" / "keyframe-interpolation" raise NotImplementedError.
    MI355X extras: two_stage_distilled=True runs the reference's DistilledPipeline class (8 steps at half resolution, x2
    upscale, 3 steps; pipelines/distilled.py:274-505), which the reference's own CLI never wires; model_version="2.3" builds
    the LTX-2.3 architecture without a checkpoint (random init, for tests and benchmarks); fp8_resident keeps fp8 checkpoint
    weights as codes in HBM; fp8_compute runs the video stream's projections fp8 x fp8 on the fp8 MFMA (BASELINE config 3; per-token /
    per-channel e4m3fn scales, not bit-identical to dequantise-at-load); vae_base_channels overrides the checkpoint's decoder_base_channels; compute_dtype="bfloat16" runs the
    bfloat16 build instead of the reference's float16 default (use_fp16=True)."""
    given = dict(upscale_temporal=upscale_temporal, early_layers_only=early_layers_only,
                 enhance_prompt_flag=enhance_prompt_flag and use_gemma, cross_attn_scale=cross_attn_scale, distilled_lora=distilled_lora,
                 stg_scale=stg_scale, apg_scale=apg_scale, control_video=control_video, save_control=save_control, ge_gamma=ge_gamma,
                 keyframes=keyframes, ic_lora_weights=ic_lora_weights, negative_prompt=negative_prompt)
    for k, v in given.items():
        if v != _OUT_OF_PATH_DEFAULTS[k]:
            raise NotImplementedError(f"{k}={v!r} is outside the MI355X hot path (see DESIGN.md); leave it at its default {_OUT_OF_PATH_DEFAULTS[k]!r}")
    if pipeline_type not in _PIPELINES_KNOWN:
        raise ValueError(f"unknown pipeline_type {pipeline_type!r}; the reference knows {_PIPELINES_KNOWN}")
    if pipeline_type not in _PIPELINES_BUILT:
        raise NotImplementedError(f"pipeline_type={pipeline_type!r} is outside the MI355X hot path (see DESIGN.md): 'two-stage' is the dev "
                                  "model's CFG stage 1 + distilled-LoRA stage 2; for the distilled two-stage DistilledPipeline pass "
                                  "two_stage_distilled=True (--two-stage-distilled)")
    output_dir = os.path.dirname(output_path)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)          # reference :1000-1003
    if num_frames % 8 != 1:
        raise ValueError(f"num_frames must be 8*k + 1, got {num_frames}")
    if height % 32 != 0 or width % 32 != 0:
        raise ValueError(f"Resolution ({height}x{width}) must be divisible by 32")
    # use_fp16=True (the reference's default, :1006): float16 operands on libltx2hip_f16.so; compute_dtype="bfloat16" (MI355X extra,
    # --bf16) selects the bfloat16 build the headline benchmark runs.  Accumulation and the residual stream are fp32 in both.
    if compute_dtype is None:
        if not use_fp16:
            raise NotImplementedError("use_fp16=False (fp32 compute): the MI355X path computes with float16 or bfloat16 operands, fp32 accumulation and an "
                                      "fp32 residual stream")
        compute_dtype = "float16"
    if compute_dtype not in ("float16", "bfloat16"):
        raise ValueError(f"compute_dtype={compute_dtype!r}: float16 or bfloat16")
    cdt = torch.float16 if compute_dtype == "float16" else torch.bfloat16
    print(f"Compute dtype: {compute_dtype} operands / fp32 accumulate / fp32 residual stream")
    if use_gemma and not (embedding_path or text_features_path):
        if not os.path.exists(gemma_path):              # the reference prints this and returns (:1085-1092)
            print(f"\n  ERROR: Gemma weights not found at {gemma_path}\n  Use use_gemma=False (--no-gemma) for dummy embeddings, or pass "
                  f"embedding_path / text_features_path")
            return None
        raise NotImplementedError("Gemma-3 text encoding is outside the hot path: pass embedding_path / text_features_path (or use_gemma=False)")
    if model_variant == "distilled" and cfg_scale > 1.2:
        print(f"  WARNING: Distilled model requires CFG=1.0 (no guidance). You requested {cfg_scale}.\n  Forcing CFG=1.0 (reference :1207-1216).")
        cfg_scale, guidance_rescale, audio_cfg_scale, rescale_scale = 1.0, 0.0, 1.0, 0.0
    # Guidance scales are resolved BEFORE any model is loaded: the AudioVideo branch's defaults for a non-distilled variant are
    # audio_cfg_scale 7.0 / rescale 0.7 (reference :1713-1716), i.e. guidance even at --cfg 1.0, and guidance needs negative encodings.
    if audio_cfg_scale is None:
        audio_cfg_scale = 1.0 if model_variant == "distilled" else 7.0
    if rescale_scale is None:
        rescale_scale = 0.0 if model_variant == "distilled" else 0.7
    negative_encoding = negative_audio_encoding = None
    if embedding_path and os.path.exists(embedding_path):
        with np.load(embedding_path) as z:                # optional entries of the pre-computed embedding file: the negative prompt's encodings
            if "negative_embedding" in z:
                negative_encoding = torch.from_numpy(z["negative_embedding"]).float()
                negative_encoding = negative_encoding[None] if negative_encoding.dim() == 2 else negative_encoding
            if "negative_audio_embedding" in z:
                negative_audio_encoding = torch.from_numpy(z["negative_audio_embedding"]).float()
                negative_audio_encoding = negative_audio_encoding[None] if negative_audio_encoding.dim() == 2 else negative_audio_encoding
    _av_branch = generate_audio or str(model_version or "").startswith("2.3") or \
        (bool(weights_path) and os.path.exists(weights_path) and model_version is None and detect_model_version(weights_path).startswith("2.3"))
    # the video-only loop of the reference guides only for cfg_scale > 1 (:1935); OneStagePipeline's guiders are enabled for any scale != 1
    _need_cfg = (cfg_scale != 1.0 or audio_cfg_scale != 1.0) if _av_branch else cfg_scale > 1.0
    if _need_cfg and not _av_branch:
        raise NotImplementedError(f"cfg_scale={cfg_scale}: classifier-free guidance is built in OneStagePipeline (the AudioVideo / LTX-2.3 branch); the "
                                  "standard video-only loop of this script runs the distilled model's cfg = 1")
    if _need_cfg and (negative_encoding is None or negative_audio_encoding is None):
        raise NotImplementedError(f"cfg_scale={cfg_scale} / audio_cfg_scale={audio_cfg_scale}: classifier-free guidance needs the NEGATIVE prompt's encodings and the text "
                                  "encoder is outside this build: add `negative_embedding` and `negative_audio_embedding` arrays to the --embedding file, or pass "
                                  "cfg_scale=1.0, audio_cfg_scale=1.0 (what --model-variant distilled does)")
    if low_memory or fast_mode:
        print("  low_memory / fast_mode: no effect here (weights and caches stay resident in HBM, the loop is one hipGraph)")
    have_ckpt = bool(weights_path) and os.path.exists(weights_path)
    version = model_version if model_version is not None else (detect_model_version(weights_path) if have_ckpt else "")
    v2 = version.startswith("2.3")                      # reference :1073: V2.3 always uses the AV transformer and the dual text encodings
    use_av_encoder = generate_audio or v2
    fps, speed = output_fps, output_speed
    torch.manual_seed(seed)
    t_all = time.time()
    base = os.path.splitext(output_path)[0]

    print("[1/5] text encoding")
    text_audio_encoding = None
    if text_features_path:
        text_encoding, _ = encode_text_features(text_features_path, weights_path if have_ckpt else None, device, seed)
    elif embedding_path:
        text_encoding, _ = load_text_embedding(embedding_path, device)
        z = np.load(embedding_path)
        if use_av_encoder and "audio_embedding" in z:
            text_audio_encoding = torch.from_numpy(z["audio_embedding"]).float().to(device)
            text_audio_encoding = text_audio_encoding[None] if text_audio_encoding.dim() == 2 else text_audio_encoding
        elif use_av_encoder:
            print("  WARNING: Pre-computed embeddings don't include audio encoding. Audio quality may be degraded.")
    else:
        # dummy encodings (reference :1129-1138).  LTX-2.3 has no caption projection: its contexts arrive at the transformer
        # widths (4096 video / 2048 audio) -- the reference's 3840-wide dummy would not fit that model at all
        text_encoding, _ = create_dummy_text_encoding(prompt, embed_dim=num_heads * 128 if v2 else 3840, device=device)
        if v2:
            text_audio_encoding, _ = create_dummy_text_encoding(prompt + " (audio)", embed_dim=num_heads * 64, device=device)
        print("  Using DUMMY encoding (test mode - output will be random)")
    if use_av_encoder and text_audio_encoding is None:
        text_audio_encoding = text_encoding             # reference :1080-1081, :1131-1132

    print("[2/5] transformer")
    model = None
    if use_placeholder:
        print("  Skipping model load (placeholder mode)")
    elif use_av_encoder:
        model = X0Model(load_av_transformer(weights_path, num_layers=num_layers, compute_dtype=cdt, use_fp8=use_fp8, low_memory=low_memory,
                                            caption_channels=None if v2 else text_encoding.shape[-1], cross_attention_adaln=v2,
                                            apply_gated_attention=v2, num_heads=num_heads, seed=seed, device=device,
                                            lora_path=lora_path, lora_strength=lora_strength, fp8_compute=fp8_compute))
    else:
        model = X0Model(load_transformer(weights_path, num_layers=num_layers, compute_dtype=cdt, use_fp8=use_fp8, low_memory=low_memory, fast_mode=fast_mode,
                                         num_heads=num_heads, caption_channels=text_encoding.shape[-1], seed=seed, device=device,
                                         lora_path=lora_path, lora_strength=lora_strength, fp8_resident=fp8_resident, fp8_compute=fp8_compute))
    print("[3/5] VAE decoder")
    vae_decoder = None
    if not skip_vae:
        vae_decoder = create_vae_decoder(weights_path, device, seed + 1, use_placeholder, vae_base_channels, compute_dtype=cdt)
    else:
        print("  VAE decoder skipped by user")
    toy = vae_base_channels is not None and vae_base_channels != 128        # debug-size VAE: matching debug-size upscaler / encoder

    def make_encoder():
        enc = SimpleVideoEncoder(device=device)
        if have_ckpt:
            load_vae_encoder_weights(enc, weights_path)
        else:
            enc.init_random_weights(seed=seed + 2)
        return enc

    def make_upscaler():
        from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler, load_spatial_upscaler_weights
        mid = 64 if toy else 1024
        if spatial_upscaler_weights == "random" or not os.path.exists(str(spatial_upscaler_weights)):
            if spatial_upscaler_weights != "random":
                print(f"  Warning: Weights not found at {spatial_upscaler_weights}, using random init")
            up = SpatialUpscaler(mid_channels=mid, num_blocks_per_stage=4 if mid == 1024 else 1, device=device)
            up.init_random_weights(seed=seed + 2)
        else:
            up = SpatialUpscaler(mid_channels=mid, device=device)
            load_spatial_upscaler_weights(up, spatial_upscaler_weights)
        return up

    def finish(frames, extra=""):
        frames_np = frames.cpu().numpy()
        np.savez_compressed(base + ".npz", frames=frames_np)
        if save_mp4:
            print(f"  video: {save_video(frames_np, output_path, fps=fps, speed=speed)}")
        print(f"Done in {time.time() - t_all:.1f} s: {base}.npz{extra}")
        return frames

    images = []
    if image_path and (two_stage_distilled or use_av_encoder):
        from ltx_2_mlx_amd.pipelines import ImageCondition
        print(f"  Image conditioning: {image_path} (strength={image_strength})")
        images = [ImageCondition(image_path=image_path, frame_index=0, strength=image_strength)]

    if two_stage_distilled:
        # MI355X extra: the reference's DistilledPipeline class (pipelines/distilled.py:274-505): 8 steps at half resolution, x2
        # latent upscale, 3 steps at full resolution, image conditioning at both resolutions, auto-tiled decode
        if not spatial_upscaler_weights:
            raise ValueError("two_stage_distilled needs --spatial-upscaler-weights (two-stage pipeline)")
        if vae_decoder is None:
            raise ValueError("the two-stage pipeline needs the VAE weights (per-channel statistics): drop --skip-vae")
        if model is None:
            raise ValueError("Two-stage pipeline requires a loaded model (cannot use placeholder mode)")
        from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
        pipe = DistilledPipeline(model, make_encoder() if images else vae_decoder, vae_decoder, spatial_upscaler=make_upscaler())
        conf = DistilledConfig(height=height, width=width, num_frames=num_frames, seed=seed, fps=24.0, use_hip_graph=use_hip_graph,
                               audio_enabled=generate_audio, tiling_config=TilingConfig.default() if tiled_vae else None)
        print("[4/5] two-stage distilled pipeline (8 steps at half resolution, x2 upscale, 3 steps)" + (" with the audio branch" if use_av_encoder else ""))
        t0 = time.time()
        out = pipe(text_encoding, None, conf, images=images, audio_encoding=text_audio_encoding)
        frames, audio_latent = out if generate_audio else (out, None)
        if audio_latent is not None:
            np.savez(base + "_audio_latent.npz", latent=audio_latent.float().cpu().numpy())     # audio VAE / vocoder are outside this path
        frames = _frames_from_video(frames)
        torch.cuda.synchronize()
        print(f"  two-stage: {(time.time() - t0):.3f} s -> {tuple(frames.shape)}")
        return finish(frames)

    if use_av_encoder:
        # === AUDIO-VIDEO PIPELINE (reference :1638-1776): OneStagePipeline on the AudioVideo transformer; LTX-2.3 always ===
        print("\n=== Using Audio-Video Pipeline ===")
        if model is None:
            print("  AV pipeline requires model - cannot use placeholder mode")
            return None
        if vae_decoder is None:
            raise ValueError("AV pipeline requires VAE decoder")
        if upscale_spatial:
            raise NotImplementedError("upscale_spatial with the AudioVideo pipeline (the reference's AV branch returns before its upscalers)")
        from ltx_2_mlx_amd.pipelines import OneStageCFGConfig, OneStagePipeline
        av_pipeline = OneStagePipeline(transformer=model, video_encoder=make_encoder() if images else None, video_decoder=vae_decoder)
        av_config = OneStageCFGConfig(
            height=height, width=width, num_frames=num_frames, seed=seed,
            fps=25.0,  # matches the upstream frame_rate for the audio latent shape (reference :1704-1712)
            num_inference_steps=num_steps, cfg_scale=cfg_scale,
            audio_cfg_scale=audio_cfg_scale, rescale_scale=rescale_scale,
            audio_enabled=generate_audio, use_hip_graph=use_hip_graph, tiling_config=TilingConfig.default() if tiled_vae else None)
        print(f"[5/5] Running audio-video generation ({num_steps} steps)...")
        t0 = time.time()
        video, audio_latent = av_pipeline(positive_encoding=text_encoding, negative_encoding=negative_encoding if _need_cfg else None, config=av_config,
                                          images=images, positive_audio_encoding=text_audio_encoding,
                                          negative_audio_encoding=negative_audio_encoding if _need_cfg else None)
        frames = _frames_from_video(video)
        torch.cuda.synchronize()
        print(f"  audio-video pipeline: {(time.time() - t0):.3f} s -> {tuple(frames.shape)}")
        if audio_latent is not None:
            np.savez(base + "_audio_latent.npz", latent=audio_latent.float().cpu().numpy())     # audio VAE / vocoder are outside this path
        return finish(frames)

    # === STANDARD PIPELINE (one-stage, distilled, video-only; reference :1778-2095) ===
    print("[4/5] latent noise")
    lf, lh, lw = (num_frames - 1) // 8 + 1, height // 32, width // 32
    g = torch.Generator(device=device).manual_seed(seed)
    latent = torch.randn(1, 128, lf, lh, lw, generator=g, device=device)
    sigmas = DISTILLED_SIGMA_VALUES[:num_steps + 1] if model_variant == "distilled" else \
        [float(s) for s in get_sigma_schedule(num_steps, distilled=False, latent=latent)]
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
        from ltx_2_mlx_amd.components import EulerDiffusionStep, GaussianNoiser
        from ltx_2_mlx_amd.pipelines import ImageCondition, apply_conditionings, create_image_conditionings, joint_denoise_loop
        tools = VideoLatentTools(patchifier, shape, fps=24.0)
        st = tools.create_initial_state(device=device)
        st = apply_conditionings(st, create_image_conditionings([ImageCondition(image_path, 0, image_strength)], make_encoder(), height, width), tools)
        st = GaussianNoiser()(st, noise_scale=1.0, noise=tok)
        st, _ = joint_denoise_loop(model, False, st, None, sigmas, text_encoding, None, EulerDiffusionStep(), None, use_hip_graph)
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
    np.savez(base + "_latent.npz", latent=latent.float().cpu().numpy())
    if upscale_spatial and spatial_upscaler_weights:
        # post-denoise 2x latent upscale (reference :2000-2037): un-normalise with the VAE statistics, upscale, re-normalise;
        # the decoded video is 2W x 2H
        from ltx_2_mlx_amd.model.upscaler import upscale_latent
        print(f"\nApplying 2x spatial upscaling...\n  Input latent: {tuple(latent.shape)}")
        up = make_upscaler()
        if vae_decoder is not None:
            stats = vae_decoder.per_channel_statistics
            latent = upscale_latent(latent, up, stats.mean_of_means, stats.std_of_means)
        else:
            print("  WARNING: No VAE decoder for normalization - output may have wrong range")
            latent = upscale_latent(latent, up, torch.zeros(128, device=device), torch.ones(128, device=device))
        print(f"  Upscaled latent: {tuple(latent.shape)}")
        np.savez(base + "_latent.npz", latent=latent.float().cpu().numpy())
    frames = None
    if vae_decoder is not None:
        t0 = time.time()
        if tiled_vae:
            frames = _frames_from_video(next(decode_tiled(latent, vae_decoder, TilingConfig.default())))
        else:
            frames = decode_latent(latent, vae_decoder)
        torch.cuda.synchronize()
        print(f"  decode: {(time.time() - t0):.3f} s -> {tuple(frames.shape)}")
        return finish(frames)
    print(f"Done in {time.time() - t_all:.1f} s: {base}_latent.npz")
    return frames


def build_parser() -> argparse.ArgumentParser:
    """Every flag of the reference's parser (scripts/generate.py:2364-2641: names, types, defaults, choices;
    tests/golden/generate_cli_flags.json), plus the MI355X extras at the end."""
    p = argparse.ArgumentParser(description="Generate video with LTX-2 on MI355X (denoise + VAE-decode hot path)")
    p.add_argument("prompt", type=str, help="Text prompt for generation")
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--width", type=int, default=704)
    p.add_argument("--frames", type=int, default=97)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--cfg", type=float, default=5.0, help="CFG scale (forced to 1.0 for the distilled model, as the reference does)")
    p.add_argument("--guidance-rescale", type=float, default=0.7)
    p.add_argument("--steps-stage1", type=int, default=15)
    p.add_argument("--steps-stage2", type=int, default=3)
    p.add_argument("--cfg-stage1", type=float, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--fps", type=int, default=24, help="output frame rate; > 24 interpolates")
    p.add_argument("--speed", type=float, default=1.0, help="playback speed multiplier")
    p.add_argument("--output", "-o", type=str, default="outputs/output.mp4")
    p.add_argument("--weights", type=str, default="weights/ltx-2/ltx-2-19b-distilled.safetensors")
    p.add_argument("--placeholder", action="store_true")
    p.add_argument("--skip-vae", action="store_true")
    p.add_argument("--embedding", type=str, default=None)
    p.add_argument("--gemma-path", type=str, default="weights/gemma-3-12b")
    p.add_argument("--no-gemma", action="store_true", help="dummy text embeddings")
    p.add_argument("--fp16", action="store_true", default=True, help="reference default: float16 operands (libltx2hip_f16.so), fp32 accumulate / residual stream; --bf16 selects the bfloat16 build")
    p.add_argument("--fp32", "--no-fp16", action="store_true", dest="fp32")
    p.add_argument("--fp8", action="store_true")
    p.add_argument("--model-variant", type=str, choices=["distilled", "dev"], default="distilled")
    p.add_argument("--distilled-lora", type=str, default=None)
    p.add_argument("--distilled-lora-scale", type=float, default=1.0)
    p.add_argument("--upscale-spatial", action="store_true")
    p.add_argument("--spatial-upscaler-weights", type=str, default="weights/ltx-2/ltx-2-spatial-upscaler-x2-1.0.safetensors")
    p.add_argument("--upscale-temporal", action="store_true")
    p.add_argument("--temporal-upscaler-weights", type=str, default="weights/ltx-2/ltx-2-temporal-upscaler-x2-1.0.safetensors")
    p.add_argument("--generate-audio", action="store_true")
    p.add_argument("--low-memory", action="store_true")
    p.add_argument("--fast-mode", action="store_true")
    p.add_argument("--image", type=str, default=None)
    p.add_argument("--image-strength", type=float, default=0.95)
    p.add_argument("--lora", type=str, default=None)
    p.add_argument("--lora-strength", type=float, default=1.0)
    p.add_argument("--stg-scale", type=float, default=0.0)
    p.add_argument("--stg-mode", type=str, choices=["video", "audio", "both"], default="video")
    p.add_argument("--apg-scale", type=float, default=1.0)
    p.add_argument("--apg-eta", type=float, default=1.0)
    p.add_argument("--apg-norm-threshold", type=float, default=0.0)
    p.add_argument("--apg-momentum", type=float, default=0.0)
    p.add_argument("--ge-gamma", type=float, default=0.0)
    p.add_argument("--control-video", type=str, default=None)
    p.add_argument("--control-type", type=str, choices=["canny", "raw"], default="raw")
    p.add_argument("--canny-low", type=int, default=100)
    p.add_argument("--canny-high", type=int, default=200)
    p.add_argument("--control-strength", type=float, default=0.95)
    p.add_argument("--save-control", action="store_true")
    p.add_argument("--tiled-vae", action="store_true")
    p.add_argument("--pipeline", type=str, choices=list(_PIPELINES_KNOWN), default="text-to-video")
    p.add_argument("--keyframe", type=str, action="append", default=None)
    p.add_argument("--ic-lora-weights", type=str, default=None)
    p.add_argument("--early-layers-only", action="store_true")
    p.add_argument("--enhance-prompt", action="store_true")
    p.add_argument("--cross-attn-scale", type=float, default=1.0)
    # --- MI355X extras (no counterpart in the reference) ---
    p.add_argument("--text-features", type=str, default=None, help="npz with Gemma `features` [T,3840] (or `hidden_states` [L,T,3840]) + `attention_mask`: run the text connector on the GPU")
    p.add_argument("--no-hip-graph", action="store_true", help="run the step loop eagerly instead of replaying the captured hipGraph")
    p.add_argument("--two-stage-distilled", action="store_true", help="DistilledPipeline: 8 steps at half resolution, x2 latent upscale (--spatial-upscaler-weights), 3 steps")
    p.add_argument("--fp8-resident", action="store_true", help="keep fp8 checkpoint weights as codes in HBM (bit-identical to dequantising at load)")
    p.add_argument("--fp8-compute", action="store_true", help="BASELINE config 3: the video stream's projections fp8 x fp8 on the CDNA4 fp8 MFMA (per-token / per-channel e4m3fn scales; "
                   "an fp8 checkpoint's codes stay resident, a 16-bit checkpoint is quantised at load); --fp8 alone keeps the reference's dequantise-at-load arithmetic")
    p.add_argument("--bf16", action="store_true", help="bfloat16 operands instead of the reference's float16 default (both: fp32 accumulate / residual stream)")
    p.add_argument("--model-version", type=str, default=None, help="force the architecture family (e.g. 2.3) instead of reading the checkpoint metadata")
    p.add_argument("--layers", type=int, default=48, help="debug: number of DiT layers for random-weight runs")
    p.add_argument("--heads", type=int, default=32, help="debug: attention heads (x128) for random-weight runs")
    p.add_argument("--vae-base-channels", type=int, default=None, help="debug: override the checkpoint's decoder_base_channels")
    p.add_argument("--no-video-file", action="store_true", help="keep only the .npz outputs (skip ffmpeg / PNG frames)")
    return p


def kwargs_from_args(a) -> dict:
    """argparse namespace -> generate_video keywords, as the reference's main() maps them (:2644-2725), incl. its weight-file
    auto-selection for --model-variant dev / --fp8."""
    if a.model_variant == "dev":
        a.weights = a.weights.replace("distilled", "dev")
        if a.steps == 7:
            a.steps = 30
    if a.fp8 and ".safetensors" in a.weights and "-fp8" not in a.weights:
        a.weights = a.weights.replace(".safetensors", "-fp8.safetensors")
    return dict(
        distilled_lora=a.distilled_lora, distilled_lora_scale=a.distilled_lora_scale, prompt=a.prompt, height=a.height, width=a.width,
        num_frames=a.frames, num_steps=a.steps, cfg_scale=a.cfg, guidance_rescale=a.guidance_rescale, seed=a.seed, weights_path=a.weights,
        output_path=a.output, use_placeholder=a.placeholder, skip_vae=a.skip_vae, embedding_path=a.embedding, gemma_path=a.gemma_path,
        use_gemma=not a.no_gemma, use_fp16=not a.fp32, use_fp8=a.fp8, model_variant=a.model_variant, upscale_spatial=a.upscale_spatial,
        spatial_upscaler_weights=a.spatial_upscaler_weights, upscale_temporal=a.upscale_temporal,
        temporal_upscaler_weights=a.temporal_upscaler_weights, generate_audio=a.generate_audio, low_memory=a.low_memory, fast_mode=a.fast_mode,
        image_path=a.image, image_strength=a.image_strength, lora_path=a.lora, lora_strength=a.lora_strength, tiled_vae=a.tiled_vae,
        pipeline_type=a.pipeline, early_layers_only=a.early_layers_only, enhance_prompt_flag=a.enhance_prompt, cross_attn_scale=a.cross_attn_scale,
        steps_stage1=a.steps_stage1, steps_stage2=a.steps_stage2, cfg_stage1=a.cfg_stage1, stg_scale=a.stg_scale, stg_mode=a.stg_mode,
        apg_scale=a.apg_scale, apg_eta=a.apg_eta, apg_norm_threshold=a.apg_norm_threshold, apg_momentum=a.apg_momentum,
        control_video=a.control_video, control_type=a.control_type, canny_low=a.canny_low, canny_high=a.canny_high,
        control_strength=a.control_strength, save_control=a.save_control, ge_gamma=a.ge_gamma, output_fps=a.fps, output_speed=a.speed,
        keyframes=a.keyframe, ic_lora_weights=a.ic_lora_weights,
        # MI355X extras
        text_features_path=a.text_features, use_hip_graph=not a.no_hip_graph, two_stage_distilled=a.two_stage_distilled,
        fp8_resident=a.fp8_resident, fp8_compute=a.fp8_compute, model_version=a.model_version, compute_dtype="bfloat16" if a.bf16 else None, num_layers=a.layers, num_heads=a.heads,
        vae_base_channels=a.vae_base_channels, save_mp4=not a.no_video_file)


def main(argv=None):
    generate_video(**kwargs_from_args(build_parser().parse_args(argv)))


if __name__ == "__main__":
    main()
