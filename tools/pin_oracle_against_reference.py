"""Generate tests/golden/*.npz by executing the REFERENCE'S OWN source files (from /root/reference)
through the throw-away mlx->torch shim (tools/mlx_shim.py), on the same seeded inputs and weights
the oracle tests regenerate.  Run in the build container only:

    python tools/pin_oracle_against_reference.py

Only outputs (and the seeds/configs that regenerate the inputs) are stored; no reference source
travels.  tests/test_oracle_golden.py checks the oracle against these vectors.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tools import mlx_shim as shim  # noqa: E402

mx, nn = shim.install()
A = shim.Arr

from oracle import dit as odit  # noqa: E402
from oracle import dit_av as oav  # noqa: E402
from oracle import upscaler as oup  # noqa: E402
from oracle import vae_encoder as oenc  # noqa: E402
from oracle import loop as oloop  # noqa: E402
from oracle import vae as ovae  # noqa: E402
from oracle import text_connector as otc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def set_param(root, dotted, tensor):
    obj = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        obj = obj[int(p)] if p.isdigit() else getattr(obj, p)
    assert hasattr(obj, parts[-1]) or parts[-1] in ("weight", "bias"), dotted
    setattr(obj, parts[-1], A(tensor.clone().float()))


def tn(x):
    return np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x.detach().float().numpy()


# ------------------------------------------------------------------------------------------ DiT
def pin_dit():
    from LTX_2_MLX.loader.weight_converter import convert_pytorch_key_to_mlx
    from LTX_2_MLX.model.transformer.model import LTXModel, LTXModelType, Modality
    from LTX_2_MLX.model.transformer.rope import LTXRopeType, precompute_freqs_cis
    from LTX_2_MLX.model.transformer.timestep_embedding import get_timestep_embedding

    cfg = odit.DiTConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=64)
    w = odit.make_dit_weights(cfg, seed=11)
    model = LTXModel(model_type=LTXModelType.VideoOnly, num_attention_heads=2, attention_head_dim=128, num_layers=2,
                     cross_attention_dim=256, caption_channels=64, compute_dtype=mx.float32)
    for k, v in w.items():
        mk = convert_pytorch_key_to_mlx(k)     # key as it is after stripping "model.diffusion_model."
        assert mk is not None, k
        set_param(model, mk, v)
    f, h, wd, S = 3, 4, 4, 16
    g = torch.Generator().manual_seed(1234)
    lat = torch.randn(1, f * h * wd, 128, generator=g)
    ctx = 0.1 * torch.randn(1, S, 64, generator=g)
    pos = oloop.video_positions(1, f, h, wd, 24.0)
    out = {}
    for tag, ts in (("scalar", torch.tensor([0.725])), ("pertoken", (torch.rand(1, f * h * wd, 1, generator=g) > 0.3).float() * 0.909375)):
        video = Modality(latent=A(lat), context=A(ctx), context_mask=None, timesteps=A(ts), positions=A(pos))
        # LTXModel.__call__ (model.py:825) passes two arguments to the VideoOnly preprocessor's
        # one-argument prepare(); call the same three stages it would run, directly.
        args = model._video_args_preprocessor.prepare(video)
        vargs, _ = model._process_transformer_blocks(args, None)
        vel = model._process_video_output(vargs.x, vargs.embedded_timestep)
        t = video.timesteps
        t = t[:, None, None] if t.ndim == 1 else t
        x0 = video.latent - t * vel
        out[f"velocity_{tag}"] = tn(vel.t)
        out[f"x0_{tag}"] = tn(x0.t)
        if tag == "scalar":
            out["adaln_emb"] = tn(args.timesteps.t)
            out["embedded_timestep"] = tn(args.embedded_timestep.t)
            out["context_proj"] = tn(args.context.t)
            out["rope_cos"] = tn(args.positional_embeddings[0].t)
            out["rope_sin"] = tn(args.positional_embeddings[1].t)
    # masked text cross-attention (attention.py:38-70, model.py:163-201): a boolean key mask (B, S) with a padded tail and a hole
    cmask = torch.ones(1, S, dtype=torch.int32)
    cmask[0, 11:] = 0
    cmask[0, 3] = 0
    ctx_m = ctx.clone()
    ctx_m[0, cmask[0] == 0] *= 40.0          # the masked keys would dominate the softmax if the mask were ignored
    video = Modality(latent=A(lat), context=A(ctx_m), context_mask=mx.array(cmask.numpy()), timesteps=A(torch.tensor([0.725])), positions=A(pos))
    args = model._video_args_preprocessor.prepare(video)
    vargs, _ = model._process_transformer_blocks(args, None)
    vel = model._process_video_output(vargs.x, vargs.embedded_timestep)
    out["x0_masked"] = tn((video.latent - video.timesteps[:, None, None] * vel).t)
    video = Modality(latent=A(lat), context=A(ctx_m), context_mask=None, timesteps=A(torch.tensor([0.725])), positions=A(pos))
    args = model._video_args_preprocessor.prepare(video)
    vargs, _ = model._process_transformer_blocks(args, None)
    out["x0_masked_control"] = tn((video.latent - video.timesteps[:, None, None] * model._process_video_output(vargs.x, vargs.embedded_timestep)).t)
    # full-width RoPE table slice + sinusoid
    pos_full = oloop.video_positions(1, 2, 3, 4, 24.0)
    c, s = precompute_freqs_cis(A(pos_full), dim=4096, out_dtype=mx.float32, theta=10000.0, max_pos=[20, 2048, 2048],
                                use_middle_indices_grid=True, num_attention_heads=32, rope_type=LTXRopeType.SPLIT)
    out["rope_full_cos"] = tn(c.t)[:, :, ::5, :]
    out["rope_full_sin"] = tn(s.t)[:, :, ::5, :]
    sig = torch.tensor(oloop.DISTILLED_SIGMA_VALUES) * 1000.0
    out["sinusoid"] = tn(get_timestep_embedding(A(sig), 256, flip_sin_to_cos=True, downscale_freq_shift=0.0).t)
    np.savez_compressed(os.path.join(GOLD, "dit_tiny.npz"), **out)
    print("dit_tiny.npz", {k: v.shape for k, v in out.items()})



# ------------------------------------------------------------------------------------------ AudioVideo DiT
def pin_dit_av():
    from LTX_2_MLX.components.patchifiers import AudioPatchifier
    from LTX_2_MLX.loader.weight_converter import convert_pytorch_key_to_mlx
    from LTX_2_MLX.model.transformer.model import LTXModel, LTXModelType, Modality, X0Model
    from LTX_2_MLX.types import AudioLatentShape

    class TinyAV(LTXModel):          # same code, 4 audio heads instead of 32 (class constants model.py:428-434)
        AUDIO_ATTENTION_HEADS = 4

    out = {}
    f, h, wd, S, Ta = 3, 4, 4, 16, 10
    for tag, v23 in (("v1", False), ("v23", True)):
        cfg = oav.AVConfig(num_attention_heads=4, attention_head_dim=128, audio_heads=4, audio_head_dim=64, num_layers=2,
                           caption_channels=None if v23 else 64, cross_attention_adaln=v23, apply_gated_attention=v23)
        w = oav.make_av_weights(cfg, seed=17 + v23)
        model = TinyAV(model_type=LTXModelType.AudioVideo, num_attention_heads=4, attention_head_dim=128, num_layers=2,
                       cross_attention_dim=512, caption_channels=cfg.caption_channels, compute_dtype=mx.float32,
                       cross_attention_adaln=v23, apply_gated_attention=v23)
        for k, v in w.items():
            mk = convert_pytorch_key_to_mlx(k, include_audio=True)
            assert mk is not None, k
            set_param(model, mk, v)
        g = torch.Generator().manual_seed(99)
        vlat = torch.randn(1, f * h * wd, 128, generator=g)
        alat = torch.randn(1, Ta, 128, generator=g)
        vctx = 0.1 * torch.randn(1, S, 64 if not v23 else cfg.inner_dim, generator=g)
        actx = 0.1 * torch.randn(1, S, 64 if not v23 else cfg.audio_inner_dim, generator=g)
        vpos = oloop.video_positions(1, f, h, wd, 24.0)
        apos_ref = AudioPatchifier(patch_size=1).get_patch_grid_bounds(AudioLatentShape(1, 8, Ta, 16))
        apos = oav.audio_positions(1, Ta)
        assert np.allclose(tn(apos_ref.t), apos.numpy()), "audio positions differ"
        sigma = 0.725
        vmask = (torch.rand(1, f * h * wd, 1, generator=g) > 0.2).float()
        for tsk, vts, ats in (("scalar", torch.tensor([sigma]), torch.tensor([sigma])),
                              ("pertoken", vmask * sigma, torch.ones(1, Ta, 1) * sigma)):
            video = Modality(latent=A(vlat), context=A(vctx), context_mask=None, timesteps=A(vts), positions=A(vpos),
                             sigma=A(torch.tensor([sigma])))
            audio = Modality(latent=A(alat), context=A(actx), context_mask=None, timesteps=A(ats), positions=A(apos),
                             sigma=A(torch.tensor([sigma])))
            vv, av = model(video, audio)
            vx0, ax0 = X0Model(model)(video, audio)
            out[f"{tag}_{tsk}_video_velocity"] = tn(vv.t)
            out[f"{tag}_{tsk}_audio_velocity"] = tn(av.t)
            out[f"{tag}_{tsk}_video_x0"] = tn(vx0.t)
            out[f"{tag}_{tsk}_audio_x0"] = tn(ax0.t)
        # video-only inference on the AudioVideo model (model.py:829-840; blocks with an empty audio stream, transformer.py:479-483)
        # under image conditioning: token 0 and most of the first latent frame are conditioned, so timesteps[0] = 0 != sigma
        cmask = torch.ones(1, f * h * wd, 1)
        cmask[:, :h * wd] = 0.05
        cmask[:, 0] = 0.0
        video = Modality(latent=A(vlat), context=A(vctx), context_mask=None, timesteps=A(cmask * sigma), positions=A(vpos),
                         sigma=A(torch.tensor([sigma])))
        res = X0Model(model)(video, None)
        vx0 = res[0] if isinstance(res, tuple) else res
        out[f"{tag}_videoonly_x0"] = tn(vx0.t)
        # masked text cross-attention of BOTH modalities (transformer.py:523, 551 hand each modality's context_mask to its attn2): different
        # boolean key masks, the masked keys' context rows x 40 so that ignoring a mask lands on the recorded control instead
        vm = torch.ones(1, S, dtype=torch.int32)
        vm[0, 10:] = 0
        vm[0, 2] = 0
        am = torch.ones(1, S, dtype=torch.int32)
        am[0, :5] = 0
        vctx_m, actx_m = vctx.clone(), actx.clone()
        vctx_m[0, vm[0] == 0] *= 40.0
        actx_m[0, am[0] == 0] *= 40.0
        ts1 = torch.tensor([sigma])
        for name, vmask_, amask_ in (("masked", vm, am), ("masked_control", None, None)):
            video = Modality(latent=A(vlat), context=A(vctx_m), context_mask=None if vmask_ is None else mx.array(vmask_.numpy()), timesteps=A(ts1),
                             positions=A(vpos), sigma=A(ts1))
            audio = Modality(latent=A(alat), context=A(actx_m), context_mask=None if amask_ is None else mx.array(amask_.numpy()), timesteps=A(ts1),
                             positions=A(apos), sigma=A(ts1))
            vx0, ax0 = X0Model(model)(video, audio)
            out[f"{tag}_{name}_video_x0"] = tn(vx0.t)
            out[f"{tag}_{name}_audio_x0"] = tn(ax0.t)
    np.savez_compressed(os.path.join(GOLD, "dit_av_tiny.npz"), **out)
    print("dit_av_tiny.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ spatial upscaler
def pin_upscaler():
    from LTX_2_MLX.model.upscaler.spatial import SpatialUpscaler
    cin, mid, nb = 64, 64, 2
    w = oup.make_upscaler_weights(cin, mid, nb, seed=41)
    up = SpatialUpscaler(in_channels=cin, mid_channels=mid, num_blocks_per_stage=nb, num_groups=32)
    up.initial_conv_weight, up.initial_conv_bias = A(w["initial_conv.weight"]), A(w["initial_conv.bias"])
    up.initial_norm.weight, up.initial_norm.bias = A(w["initial_norm.weight"]), A(w["initial_norm.bias"])
    up.final_conv_weight, up.final_conv_bias = A(w["final_conv.weight"]), A(w["final_conv.bias"])
    for stage, blocks in (("res_blocks", up.res_blocks), ("post_upsample_res_blocks", up.post_upsample_res_blocks)):
        for i, blk in enumerate(blocks):
            blk.conv1_weight, blk.conv1_bias = A(w[f"{stage}.{i}.conv1.weight"]), A(w[f"{stage}.{i}.conv1.bias"])
            blk.conv2_weight, blk.conv2_bias = A(w[f"{stage}.{i}.conv2.weight"]), A(w[f"{stage}.{i}.conv2.bias"])
            blk.norm1.weight, blk.norm1.bias = A(w[f"{stage}.{i}.norm1.weight"]), A(w[f"{stage}.{i}.norm1.bias"])
            blk.norm2.weight, blk.norm2.bias = A(w[f"{stage}.{i}.norm2.weight"]), A(w[f"{stage}.{i}.norm2.bias"])
    # PyTorch conv2d weight (out, in, kh, kw) -> MLX (out, kh, kw, in), as _load_upsampler_weight does (spatial.py:520-531)
    up.upsampler.conv_weight = A(w["upsampler.conv.weight"].permute(0, 2, 3, 1).contiguous())
    up.upsampler.conv_bias = A(w["upsampler.conv.bias"])
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, cin, 3, 5, 6, generator=g)
    out = {"upscaled": tn(up(A(x)).t)}
    np.savez_compressed(os.path.join(GOLD, "upscaler_tiny.npz"), **out)
    print("upscaler_tiny.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ text connector / feature extractors
def pin_text_connector():
    """Embeddings1DConnector (2 heads x 128, 2 layers, 16 registers -> 1024 tokens) on a 40-token prompt, with the
    fp32 and the float64 frequency grid; both Gemma feature extractors on a left-padded 2 x 12 x 32 x 5 stack."""
    from LTX_2_MLX.model.text_encoder.connector import Embeddings1DConnector
    from LTX_2_MLX.model.text_encoder.feature_extractor import GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2
    import LTX_2_MLX.model.transformer.rope as ref_rope
    from LTX_2_MLX.model.transformer.rope import LTXRopeType
    ref_rope._HAS_FUSED_ROPE = False           # the fused variant is a Metal kernel; the reference's own fallback is the definition
    out = {}
    for tag, dbl in (("f32", False), ("f64", True)):
        cfg = otc.ConnectorConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, num_learnable_registers=16,
                                  double_precision_rope=dbl)
        w = otc.make_connector_weights(cfg, seed=61)
        conn = Embeddings1DConnector(attention_head_dim=128, num_attention_heads=2, num_layers=2, num_learnable_registers=16,
                                     rope_type=LTXRopeType.INTERLEAVED, double_precision_rope=dbl)
        conn.learnable_registers = A(w["learnable_registers"])
        for i, blk in enumerate(conn.transformer_1d_blocks):
            p = f"transformer_1d_blocks.{i}"
            for n, attr in (("to_q", "to_q"), ("to_k", "to_k"), ("to_v", "to_v"), ("to_out.0", "to_out")):
                lin = getattr(blk.attn1, attr)
                lin.weight, lin.bias = A(w[f"{p}.attn1.{n}.weight"]), A(w[f"{p}.attn1.{n}.bias"])
            blk.attn1.q_norm.weight, blk.attn1.k_norm.weight = A(w[f"{p}.attn1.q_norm.weight"]), A(w[f"{p}.attn1.k_norm.weight"])
            blk.ff.project_in.proj.weight, blk.ff.project_in.proj.bias = A(w[f"{p}.ff.net.0.proj.weight"]), A(w[f"{p}.ff.net.0.proj.bias"])
            blk.ff.project_out.weight, blk.ff.project_out.bias = A(w[f"{p}.ff.net.2.weight"]), A(w[f"{p}.ff.net.2.bias"])
        g = torch.Generator().manual_seed(62)
        x = torch.randn(1, 40, cfg.inner_dim, generator=g)
        add_mask = torch.zeros(1, 1, 1, 40)
        add_mask[..., 30:] = -3.4e38          # ten pad tokens: the connector must clear the mask
        y, m = conn(A(x), A(add_mask))
        yt = tn(y.t)                           # keep the prompt rows, the first registers and the last rows (+ global stats)
        out[f"connector_{tag}_head"], out[f"connector_{tag}_tail"] = yt[:, :56], yt[:, 992:]
        out[f"connector_{tag}_stats"] = np.array([yt.mean(), yt.std(), np.abs(yt).max()], dtype=np.float32)
        assert float(np.abs(tn(m.t)).max()) == 0.0 and y.t.shape == (1, 1024, cfg.inner_dim)
    g = torch.Generator().manual_seed(63)
    hs = [torch.randn(2, 12, 32, generator=g) * (1 + 0.3 * i) + 0.1 * i for i in range(5)]
    am = torch.ones(2, 12)
    am[1, :5] = 0                              # left padding, 7 valid tokens
    fe1 = GemmaFeaturesExtractorProjLinear(hidden_dim=32, num_layers=5)
    w1 = 0.05 * torch.randn(32, 160, generator=g)
    fe1.aggregate_embed.weight = A(w1)
    out["fe_v1_left"] = tn(fe1.extract_from_hidden_states([A(h) for h in hs], A(am), padding_side="left").t)
    am_r = torch.ones(2, 12)
    am_r[1, 7:] = 0
    out["fe_v1_right"] = tn(fe1.extract_from_hidden_states([A(h) for h in hs], A(am_r), padding_side="right").t)
    fe2 = GemmaFeaturesExtractorV2(hidden_dim=32, num_layers=5, video_inner_dim=48, audio_inner_dim=24)
    wv, bv = 0.05 * torch.randn(48, 160, generator=g), 0.1 * torch.randn(48, generator=g)
    wa, ba = 0.05 * torch.randn(24, 160, generator=g), 0.1 * torch.randn(24, generator=g)
    fe2.video_aggregate_embed.weight, fe2.video_aggregate_embed.bias = A(wv), A(bv)
    fe2.audio_aggregate_embed.weight, fe2.audio_aggregate_embed.bias = A(wa), A(ba)
    v, a = fe2.extract_from_hidden_states([A(h) for h in hs], A(am))
    out["fe_v2_video"], out["fe_v2_audio"] = tn(v.t), tn(a.t)
    # the oracle on the same inputs, here and again (from the seeds) in tests/test_oracle_golden.py
    chk = otc.feature_extractor_v1(hs, am, {"aggregate_embed.weight": w1}, "left")
    print("  fe_v1 oracle vs reference:", float((chk - torch.from_numpy(out["fe_v1_left"])).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "text_connector.npz"), **out)
    print("text_connector.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ VAE encoder + conditioning
def pin_vae_encoder():
    """The reference hard-wires the encoder widths (128..1024, 28 res blocks), so the full-size encoder runs on
    tiny inputs: one 64x64 image (image conditioning) and a 9-frame 64x96 clip (temporal downsampling)."""
    from LTX_2_MLX.components.patchifiers import VideoLatentPatchifier
    from LTX_2_MLX.conditioning.latent import VideoConditionByLatentIndex
    from LTX_2_MLX.conditioning.tools import VideoLatentTools
    from LTX_2_MLX.model.video_vae.simple_encoder import SimpleVideoEncoder
    from LTX_2_MLX.types import VideoLatentShape
    w = oenc.make_encoder_weights(seed=51)
    enc = SimpleVideoEncoder(compute_dtype=mx.float32)
    enc.per_channel_statistics.mean_of_means = A(w["vae.per_channel_statistics.mean-of-means"])
    enc.per_channel_statistics.std_of_means = A(w["vae.per_channel_statistics.std-of-means"])
    for sfx in ("weight", "bias"):
        setattr(enc.conv_in, sfx, A(w[f"vae.encoder.conv_in.conv.{sfx}"]))
        setattr(enc.conv_out, sfx, A(w[f"vae.encoder.conv_out.conv.{sfx}"]))
    for i, (kind, arg) in enumerate(oenc.DEFAULT_BLOCKS):
        blk = getattr(enc, f"down_blocks_{i}")
        for sfx in ("weight", "bias"):
            if kind == "res":
                for j, rb in enumerate(blk.res_blocks):
                    setattr(rb.conv1, sfx, A(w[f"vae.encoder.down_blocks.{i}.res_blocks.{j}.conv1.conv.{sfx}"]))
                    setattr(rb.conv2, sfx, A(w[f"vae.encoder.down_blocks.{i}.res_blocks.{j}.conv2.conv.{sfx}"]))
            else:
                setattr(blk.conv, sfx, A(w[f"vae.encoder.down_blocks.{i}.conv.conv.{sfx}"]))
    g = torch.Generator().manual_seed(52)
    img = torch.rand(1, 3, 1, 64, 64, generator=g) * 2 - 1
    clip = torch.rand(1, 3, 9, 64, 96, generator=g) * 2 - 1
    out = {"image_latent": tn(enc(A(img), show_progress=False).t), "clip_latent": tn(enc(A(clip), show_progress=False).t)}
    # VideoConditionByLatentIndex on a 3x2x2 latent grid, strength 0.8 at latent frame 0
    shp = VideoLatentShape(batch=1, channels=128, frames=3, height=2, width=2)
    tools = VideoLatentTools(patchifier=VideoLatentPatchifier(patch_size=1), target_shape=shp, fps=24.0)
    st = tools.create_initial_state()
    cond = VideoConditionByLatentIndex(latent=A(torch.from_numpy(out["image_latent"])), strength=0.8, latent_idx=0)
    st2 = cond.apply_to(st, tools)
    out["cond_latent"], out["cond_mask"], out["cond_clean"] = tn(st2.latent.t), tn(st2.denoise_mask.t), tn(st2.clean_latent.t)
    np.savez_compressed(os.path.join(GOLD, "vae_encoder.npz"), **out)
    print("vae_encoder.npz", {k: v.shape for k, v in out.items()})

# ------------------------------------------------------------------------------------------ loop helpers
def pin_loop():
    from LTX_2_MLX.components.diffusion_steps import EulerDiffusionStep
    from LTX_2_MLX.components.patchifiers import VideoLatentPatchifier, get_pixel_coords
    from LTX_2_MLX.components.schedulers import DISTILLED_SIGMA_VALUES, STAGE_2_DISTILLED_SIGMA_VALUES, LTX2Scheduler
    from LTX_2_MLX.conditioning.tools import VideoLatentTools
    from LTX_2_MLX.pipelines.common import post_process_latent, timesteps_from_mask
    from LTX_2_MLX.types import VideoLatentShape

    out = {"distilled": np.array(DISTILLED_SIGMA_VALUES), "stage2": np.array(STAGE_2_DISTILLED_SIGMA_VALUES)}
    for steps in (2, 8, 30):
        out[f"ltx2_sched_{steps}"] = tn(LTX2Scheduler().execute(steps).t)
    out["ltx2_sched_8_tokens3456"] = tn(LTX2Scheduler().execute(8, latent=mx.zeros((1, 128, 9, 16, 24))).t)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 24, 128, generator=g)
    x0 = torch.randn(1, 24, 128, generator=g)
    out["euler"] = tn(EulerDiffusionStep().step(A(x), A(x0), A(torch.tensor(DISTILLED_SIGMA_VALUES)), 5).t)
    mask = (torch.rand(1, 24, 1, generator=g) > 0.5).float()
    out["post_process"] = tn(post_process_latent(A(x0), A(mask), A(x)).t)
    out["timesteps_from_mask"] = tn(timesteps_from_mask(A(mask), 0.725).t)
    shp = VideoLatentShape(batch=1, channels=128, frames=3, height=4, width=5)
    st = VideoLatentTools(patchifier=VideoLatentPatchifier(patch_size=1), target_shape=shp, fps=24.0).create_initial_state()
    out["positions_3x4x5_fps24"] = tn(st.positions.t)
    lat5 = torch.randn(1, 128, 3, 4, 5, generator=g)
    out["patchify"] = tn(VideoLatentPatchifier(patch_size=1).patchify(A(lat5)).t)
    # audio side of the loop helpers (AudioVideo pipelines)
    from LTX_2_MLX.components.patchifiers import AudioPatchifier
    from LTX_2_MLX.conditioning.tools import AudioLatentTools
    from LTX_2_MLX.pipelines.distilled import DistilledPipeline
    from LTX_2_MLX.types import AudioLatentShape, VideoPixelShape
    ashape = AudioLatentShape.from_video_pixel_shape(VideoPixelShape(batch=1, frames=65, height=512, width=768, fps=24.0))
    out["audio_shape_65f_24fps"] = np.array(ashape.to_tuple())
    ast = AudioLatentTools(patchifier=AudioPatchifier(patch_size=1), target_shape=AudioLatentShape(1, 8, 11, 16)).create_initial_state()
    out["audio_positions_11"] = tn(ast.positions.t)
    alat = torch.randn(1, 8, 11, 16, generator=g)
    out["audio_patchify"] = tn(AudioPatchifier(patch_size=1).patchify(A(alat)).t)
    anoise = torch.randn(1, 37, 128, generator=g) * 1.7 + 0.3
    out["audio_channelwise_normalize"] = tn(DistilledPipeline._channelwise_normalize_audio(A(anoise)).t)
    np.savez_compressed(os.path.join(GOLD, "loop.npz"), **out)
    print("loop.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ VAE
def load_ref_decoder(dec, w, cfg):
    from LTX_2_MLX.model.video_vae.simple_decoder import TimestepEmbedder
    dec.mean_of_means = A(w["vae.per_channel_statistics.mean-of-means"])
    dec.std_of_means = A(w["vae.per_channel_statistics.std-of-means"])
    for sfx in ("weight", "bias"):
        setattr(dec.conv_in, sfx, A(w[f"vae.decoder.conv_in.conv.{sfx}"]))
        setattr(dec.conv_out, sfx, A(w[f"vae.decoder.conv_out.conv.{sfx}"]))
    for i, (block, btype) in enumerate(zip(dec.up_blocks, dec.block_types)):
        pre = f"vae.decoder.up_blocks.{i}"
        if btype == "res":
            for j, rb in enumerate(block.res_blocks):
                for cn in ("conv1", "conv2"):
                    for sfx in ("weight", "bias"):
                        setattr(getattr(rb, cn), sfx, A(w[f"{pre}.res_blocks.{j}.{cn}.conv.{sfx}"]))
                rb.scale_shift_table = A(w[f"{pre}.res_blocks.{j}.scale_shift_table"])
            k1 = f"{pre}.time_embedder.timestep_embedder.linear_1.weight"
            if k1 in w:
                te = TimestepEmbedder(hidden_dim=w[k1].shape[0], output_dim=4 * block.channels, input_dim=256)
                for ln in ("linear_1", "linear_2"):
                    for sfx in ("weight", "bias"):
                        setattr(getattr(te, ln), sfx, A(w[f"{pre}.time_embedder.timestep_embedder.{ln}.{sfx}"]))
                block.time_embedder = te
        else:
            for sfx in ("weight", "bias"):
                setattr(block.conv, sfx, A(w[f"{pre}.conv.conv.{sfx}"]))
    dec.last_scale_shift_table = A(w["vae.decoder.last_scale_shift_table"])
    if cfg.timestep_conditioning:
        dec.timestep_scale_multiplier = A(w["vae.decoder.timestep_scale_multiplier"])
        lt = "vae.decoder.last_time_embedder.timestep_embedder"
        te = TimestepEmbedder(hidden_dim=256, output_dim=2 * dec.final_channels, input_dim=256)
        for ln in ("linear_1", "linear_2"):
            for sfx in ("weight", "bias"):
                setattr(getattr(te, ln), sfx, A(w[f"{lt}.{ln}.{sfx}"]))
        dec.last_time_embedder = te


def pin_vae():
    from LTX_2_MLX.model.video_vae.ops import unpatchify
    from LTX_2_MLX.model.video_vae.simple_decoder import (Conv3dSimple, DepthToSpaceUpsample3d, SimpleVideoDecoder,
                                                          decode_latent)
    from LTX_2_MLX.model.video_vae.tiling import (SpatialTilingConfig, TemporalTilingConfig, TilingConfig,
                                                  compute_trapezoidal_mask_1d, decode_tiled, generate_tile_specs)
    out = {}
    g = torch.Generator().manual_seed(21)
    # single conv (causal / non-causal) and the three depth-to-space variants
    x = torch.randn(1, 8, 3, 5, 6, generator=g)
    wc = torch.randn(16, 8, 3, 3, 3, generator=g) / 15.0
    bc = torch.randn(16, generator=g)
    conv = Conv3dSimple(8, 16)
    conv.weight, conv.bias = A(wc), A(bc)
    out["conv_noncausal"] = tn(conv(A(x), causal=False).t)
    out["conv_causal"] = tn(conv(A(x), causal=True).t)
    for name, stride, mult, resid in (("all", (2, 2, 2), 2, True), ("space", (1, 2, 2), 2, True), ("time", (2, 1, 1), 1, False)):
        cin = 16
        sp = stride[0] * stride[1] * stride[2]
        up = DepthToSpaceUpsample3d(cin, stride=stride, residual=resid, out_channels_reduction_factor=mult)
        wu = torch.randn(sp * cin // mult, cin, 3, 3, 3, generator=g) / 20.0
        bu = torch.randn(sp * cin // mult, generator=g)
        up.conv.weight, up.conv.bias = A(wu), A(bu)
        xu = torch.randn(1, cin, 2, 3, 4, generator=g)
        out[f"up_{name}"] = tn(up(A(xu), causal=False).t)
    xp = torch.randn(1, 48, 2, 3, 4, generator=g)
    out["unpatchify"] = tn(unpatchify(A(xp), patch_size_hw=4, patch_size_t=1).t)

    # tiny full decoder, default block structure, with timestep conditioning and a shared noise tensor
    blocks = [["res_x", {"num_layers": 2}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}]]
    cfg = ovae.VAEConfig(decoder_blocks=blocks, base_channels=8, timestep_conditioning=True)
    w = ovae.make_vae_weights(cfg, seed=31)
    dec = SimpleVideoDecoder(decoder_blocks=blocks, base_channels=8, timestep_conditioning=True, compute_dtype=mx.float32)
    load_ref_decoder(dec, w, cfg)
    z = torch.randn(1, 128, 2, 2, 3, generator=g)
    nz = torch.randn(1, 128, 2, 2, 3, generator=g)
    shim._Random.preset.append(nz)
    out["decoder_tcond"] = tn(dec(A(z), timestep=0.05, show_progress=False).t)[..., ::2, ::2]
    out["decoder_no_t"] = tn(dec(A(z), timestep=None, show_progress=False).t)[..., ::2, ::2]
    # chunked decode_latent at T'=9 (noise per chunk shared with the oracle through `preset`)
    z9 = torch.randn(1, 128, 9, 2, 2, generator=g)
    n9 = torch.randn(1, 128, 9, 2, 2, generator=g)
    for s, e in ovae.temporal_chunks(9):
        shim._Random.preset.append(n9[:, :, s:e].contiguous())
    import builtins
    _print = builtins.print
    frames = decode_latent(A(z9), dec, timestep=0.05)
    out["decode_latent_u8"] = np.asarray(frames.t.numpy(), dtype=np.uint8)[::2, ::2, ::2]
    # tiled decode (no timestep conditioning -> deterministic)
    zt = torch.randn(1, 128, 4, 4, 6, generator=g)
    tc = TilingConfig(SpatialTilingConfig(96, 32), TemporalTilingConfig(16, 8))
    tiled = next(decode_tiled(A(zt), lambda lt, timestep=None: dec(lt, timestep=None, show_progress=False), tc, timestep=None,
                              show_progress=False))
    out["decode_tiled"] = tn(tiled.t)[:, :, :, ::4, ::4]
    specs = generate_tile_specs((1, 128, 9, 32, 48), TilingConfig.default())
    out["tile_specs_9x32x48"] = np.array([[s.in_t_start, s.in_t_end, s.in_h_start, s.in_h_end, s.in_w_start, s.in_w_end,
                                           s.out_t_start, s.out_t_end, s.out_h_start, s.out_h_end, s.out_w_start, s.out_w_end,
                                           s.ramp_t_left, s.ramp_t_right, s.ramp_h_left, s.ramp_h_right, s.ramp_w_left, s.ramp_w_right]
                                          for s in specs])
    out["trapezoid_10_3_2"] = tn(compute_trapezoidal_mask_1d(10, 3, 2, False).t)
    out["trapezoid_64_0_24_from0"] = tn(compute_trapezoidal_mask_1d(64, 0, 24, True).t)
    np.savez_compressed(os.path.join(GOLD, "vae_tiny.npz"), **out)
    print("vae_tiny.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------ guiders
def pin_guiders():
    """CFGGuider / CFGStarRescalingGuider / projection_coef of the reference (components/guiders.py:26-77, 290-306) on seeded tensors."""
    from LTX_2_MLX.components.guiders import CFGGuider, CFGStarRescalingGuider, projection_coef
    g = torch.Generator().manual_seed(4242)
    cond = torch.randn(1, 48, 128, generator=g)
    uncond = 0.7 * cond + 0.5 * torch.randn(1, 48, 128, generator=g)
    out = {"projection_coef": tn(projection_coef(A(cond), A(uncond)).t)}
    for sc in (1.0, 3.0, 7.0):
        out[f"cfg_{sc}"] = tn(CFGGuider(scale=sc).guide(A(cond), A(uncond)).t)
        out[f"cfgstar_{sc}"] = tn(CFGStarRescalingGuider(scale=sc).guide(A(cond), A(uncond)).t)
    assert not CFGGuider(scale=1.0).enabled() and CFGStarRescalingGuider(scale=3.0).enabled()
    np.savez_compressed(os.path.join(GOLD, "guiders.npz"), **out)
    print("guiders.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "guiders":
        with torch.no_grad():
            pin_guiders()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "text_connector":
        pin_text_connector()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dit":
        with torch.no_grad():
            pin_dit()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dit_av":
        with torch.no_grad():
            pin_dit_av()
        sys.exit(0)
    with torch.no_grad():
        pin_loop()
        pin_dit()
        pin_dit_av()
        pin_upscaler()
        pin_vae_encoder()
        pin_vae()
        pin_guiders()
    print("golden vectors written to", GOLD)
