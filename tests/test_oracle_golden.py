"""The oracle against golden vectors recorded from the REFERENCE'S OWN source files
(tools/pin_oracle_against_reference.py: reference modules from /root/reference executed over a
throw-away mlx->torch leaf-op shim in the build container).  Inputs and weights are regenerated
here from the same seeds; only reference outputs live in tests/golden/*.npz.  CPU-only."""
import os

import numpy as np
import pytest
import torch

from oracle import dit, loop, vae

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def g(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().float().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_loop_helpers_match_reference():
    z = g("loop.npz")
    assert loop.DISTILLED_SIGMA_VALUES == list(z["distilled"]) and loop.STAGE_2_DISTILLED_SIGMA_VALUES == list(z["stage2"])
    for steps in (2, 8, 30):
        close(loop.ltx2_scheduler(steps), z[f"ltx2_sched_{steps}"], rtol=1e-5, atol=1e-6)
    close(loop.ltx2_scheduler(8, tokens=3456), z["ltx2_sched_8_tokens3456"], rtol=1e-5, atol=1e-6)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 24, 128, generator=gen)
    x0 = torch.randn(1, 24, 128, generator=gen)
    sig = loop.DISTILLED_SIGMA_VALUES
    close(loop.euler_step(x, x0, sig[5], sig[6]), z["euler"], rtol=1e-5, atol=1e-6)
    mask = (torch.rand(1, 24, 1, generator=gen) > 0.5).float()
    close(loop.post_process_latent(x0, mask, x), z["post_process"], rtol=0, atol=0)
    close(loop.timesteps_from_mask(mask, 0.725), z["timesteps_from_mask"], rtol=1e-6, atol=0)
    close(loop.video_positions(1, 3, 4, 5, 24.0), z["positions_3x4x5_fps24"], rtol=1e-6, atol=1e-7)
    lat5 = torch.randn(1, 128, 3, 4, 5, generator=gen)
    close(loop.patchify(lat5), z["patchify"], rtol=0, atol=0)


def test_product_host_logic_matches_reference():
    """The product's own host-side mirrors (schedulers, positions, patchify) against the same vectors."""
    from ltx_2_mlx_amd.components import LTX2Scheduler, VideoLatentPatchifier
    from ltx_2_mlx_amd.conditioning import VideoLatentTools
    from ltx_2_mlx_amd.pipelines import post_process_latent
    from ltx_2_mlx_amd.types import VideoLatentShape
    z = g("loop.npz")
    for steps in (2, 8, 30):
        close(LTX2Scheduler().execute(steps), z[f"ltx2_sched_{steps}"], rtol=1e-5, atol=1e-6)
    close(LTX2Scheduler().execute(8, latent=torch.zeros(1, 128, 9, 16, 24)), z["ltx2_sched_8_tokens3456"], rtol=1e-5, atol=1e-6)
    shp = VideoLatentShape(1, 128, 3, 4, 5)
    st = VideoLatentTools(VideoLatentPatchifier(1), shp, fps=24.0).create_initial_state()
    close(st.positions, z["positions_3x4x5_fps24"], rtol=1e-6, atol=1e-7)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 24, 128, generator=gen)
    x0 = torch.randn(1, 24, 128, generator=gen)
    mask = (torch.rand(1, 24, 1, generator=gen) > 0.5).float()
    close(post_process_latent(x0, mask, x), z["post_process"], rtol=0, atol=0)
    lat5 = torch.randn(1, 128, 3, 4, 5, generator=gen)
    close(VideoLatentPatchifier(1).patchify(lat5), z["patchify"], rtol=0, atol=0)


def masked_case(ctx):
    """The masked-cross-attention case of pin_dit: padded tail + one hole; the masked keys' context rows scaled by 40."""
    cmask = torch.ones(1, ctx.shape[1], dtype=torch.int32)
    cmask[0, 11:] = 0
    cmask[0, 3] = 0
    ctx_m = ctx.clone()
    ctx_m[0, cmask[0] == 0] *= 40.0
    return cmask, ctx_m


def test_dit_matches_reference():
    z = g("dit_tiny.npz")
    cfg = dit.DiTConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=64)
    w = dit.make_dit_weights(cfg, seed=11)
    f, h, wd, S = 3, 4, 4, 16
    gen = torch.Generator().manual_seed(1234)
    lat = torch.randn(1, f * h * wd, 128, generator=gen)
    ctx = 0.1 * torch.randn(1, S, 64, generator=gen)
    pos = loop.video_positions(1, f, h, wd, 24.0)
    ts = torch.tensor([0.725])
    emb, e = dit.prepare_timestep(ts.reshape(1, -1), w, cfg, 1)
    close(emb, z["adaln_emb"], rtol=2e-4, atol=2e-5)
    close(e, z["embedded_timestep"], rtol=2e-4, atol=2e-5)
    close(dit.caption_projection(ctx, w), z["context_proj"], rtol=2e-4, atol=2e-5)
    cos, sin = dit.rope_split_tables(pos, cfg.inner_dim, 2, 10000.0, [20, 2048, 2048])
    close(cos, z["rope_cos"], rtol=0, atol=2e-5)
    close(sin, z["rope_sin"], rtol=0, atol=2e-5)
    close(dit.velocity_model(lat, ctx, ts, pos, w, cfg), z["velocity_scalar"], rtol=2e-3, atol=2e-4)
    close(dit.x0_model(lat, ctx, ts, pos, w, cfg), z["x0_scalar"], rtol=2e-3, atol=2e-4)
    tsp = (torch.rand(1, f * h * wd, 1, generator=gen) > 0.3).float() * 0.909375
    close(dit.velocity_model(lat, ctx, tsp, pos, w, cfg), z["velocity_pertoken"], rtol=2e-3, atol=2e-4)
    close(dit.x0_model(lat, ctx, tsp, pos, w, cfg), z["x0_pertoken"], rtol=2e-3, atol=2e-4)
    # masked text cross-attention: boolean key mask -> additive -3.4e38 (model.py:163-201, attention.py:38-70); the masked keys carry
    # 40x larger context rows, so ignoring the mask lands on the recorded control instead
    cmask, ctx_m = masked_case(ctx)
    xm = dit.x0_model(lat, ctx_m, ts, pos, w, cfg, context_mask=cmask)
    close(xm, z["x0_masked"], rtol=2e-3, atol=2e-4)
    close(dit.x0_model(lat, ctx_m, ts, pos, w, cfg), z["x0_masked_control"], rtol=2e-3, atol=2e-4)
    assert float(np.abs(z["x0_masked"] - z["x0_masked_control"]).max()) > 5e-3
    # full-width RoPE (dim 4096, 32 heads) and the timestep sinusoid at the distilled sigmas
    posf = loop.video_positions(1, 2, 3, 4, 24.0)
    cf, sf = dit.rope_split_tables(posf, 4096, 32, 10000.0, [20, 2048, 2048])
    close(cf[:, :, ::5], z["rope_full_cos"], rtol=0, atol=2e-3)      # arguments reach ~1.6e4 rad in fp32
    close(sf[:, :, ::5], z["rope_full_sin"], rtol=0, atol=2e-3)
    close(dit.sinusoidal_timestep_embedding(torch.tensor(loop.DISTILLED_SIGMA_VALUES) * 1000.0), z["sinusoid"], rtol=0, atol=2e-4)


def test_product_rope_tables_match_reference():
    from ltx_2_mlx_amd.model.transformer import rope_tables_token_major
    z = g("dit_tiny.npz")
    posf = loop.video_positions(1, 2, 3, 4, 24.0)
    c, s = rope_tables_token_major(posf, 4096, 32, 10000.0, [20, 2048, 2048])
    ref_c = torch.from_numpy(z["rope_full_cos"])[0].permute(1, 0, 2).reshape(-1, 2048)      # tokens 0,5,10,15,20
    ref_s = torch.from_numpy(z["rope_full_sin"])[0].permute(1, 0, 2).reshape(-1, 2048)
    close(c[::5], ref_c.numpy(), rtol=0, atol=2e-3)
    close(s[::5], ref_s.numpy(), rtol=0, atol=2e-3)


def _vae_ops():
    z = g("vae_tiny.npz")
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(1, 8, 3, 5, 6, generator=gen)
    wc = torch.randn(16, 8, 3, 3, 3, generator=gen) / 15.0
    bc = torch.randn(16, generator=gen)
    close(vae.conv3d_simple(x, wc, bc, causal=False), z["conv_noncausal"], rtol=1e-4, atol=1e-5)
    close(vae.conv3d_simple(x, wc, bc, causal=True), z["conv_causal"], rtol=1e-4, atol=1e-5)
    for name, stride, mult, resid in (("all", (2, 2, 2), 2, True), ("space", (1, 2, 2), 2, True), ("time", (2, 1, 1), 1, False)):
        cin = 16
        sp = stride[0] * stride[1] * stride[2]
        wu = torch.randn(sp * cin // mult, cin, 3, 3, 3, generator=gen) / 20.0
        bu = torch.randn(sp * cin // mult, generator=gen)
        xu = torch.randn(1, cin, 2, 3, 4, generator=gen)
        out = vae.upsample_block(xu, {"u.conv.conv.weight": wu, "u.conv.conv.bias": bu}, "u", stride, mult, resid, causal=False)
        close(out, z[f"up_{name}"], rtol=1e-4, atol=1e-5)
    xp = torch.randn(1, 48, 2, 3, 4, generator=gen)
    close(vae.unpatchify(xp, 4, 1), z["unpatchify"], rtol=0, atol=0)
    return gen


def test_vae_ops_match_reference():
    _vae_ops()


def test_vae_decoder_and_decode_paths_match_reference():
    z = g("vae_tiny.npz")
    gen = _vae_ops()          # advances the generator exactly like the recording script
    blocks = [["res_x", {"num_layers": 2}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": 1}]]
    cfg = vae.VAEConfig(decoder_blocks=blocks, base_channels=8, timestep_conditioning=True)
    w = vae.make_vae_weights(cfg, seed=31)
    lat = torch.randn(1, 128, 2, 2, 3, generator=gen)
    nz = torch.randn(1, 128, 2, 2, 3, generator=gen)
    close(vae.decoder_forward(lat, w, cfg, timestep=0.05, noise=nz)[..., ::2, ::2], z["decoder_tcond"], rtol=2e-3, atol=2e-4)
    close(vae.decoder_forward(lat, w, cfg, timestep=None)[..., ::2, ::2], z["decoder_no_t"], rtol=2e-3, atol=2e-4)
    z9 = torch.randn(1, 128, 9, 2, 2, generator=gen)
    n9 = torch.randn(1, 128, 9, 2, 2, generator=gen)
    frames = vae.decode_latent(z9, w, cfg, timestep=0.05, noise=n9)
    assert frames.shape == (65, 64, 64, 3)
    d = np.abs(frames[::2, ::2, ::2].numpy().astype(np.int32) - z["decode_latent_u8"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3           # truncation ties only
    zt = torch.randn(1, 128, 4, 4, 6, generator=gen)
    tiled = vae.decode_tiled(zt, lambda t: vae.decoder_forward(t, w, cfg, timestep=None), spatial=(96, 32), temporal=(16, 8))
    close(tiled[:, :, :, ::4, ::4], z["decode_tiled"], rtol=2e-3, atol=2e-4)
    specs = vae.tile_specs((1, 128, 9, 32, 48))
    ref = z["tile_specs_9x32x48"]
    got = np.array([[*s["in_t"], *s["in_h"], *s["in_w"], *s["out_t"], *s["out_h"], *s["out_w"], *s["ramp_t"], *s["ramp_h"], *s["ramp_w"]] for s in specs])
    assert np.array_equal(got, ref)
    close(vae.trapezoid_mask_1d(10, 3, 2, False), z["trapezoid_10_3_2"], rtol=1e-6, atol=1e-7)
    close(vae.trapezoid_mask_1d(64, 0, 24, True), z["trapezoid_64_0_24_from0"], rtol=1e-6, atol=1e-7)


def av_tiny_case(v23: bool):
    """Inputs of tools/pin_oracle_against_reference.py:pin_dit_av, regenerated from the same seeds."""
    from oracle import dit_av
    cfg = dit_av.AVConfig(num_attention_heads=4, attention_head_dim=128, audio_heads=4, audio_head_dim=64, num_layers=2,
                          caption_channels=None if v23 else 64, cross_attention_adaln=v23, apply_gated_attention=v23)
    w = dit_av.make_av_weights(cfg, seed=17 + v23)
    f, h, wd, S, Ta = 3, 4, 4, 16, 10
    gen = torch.Generator().manual_seed(99)
    vlat = torch.randn(1, f * h * wd, 128, generator=gen)
    alat = torch.randn(1, Ta, 128, generator=gen)
    vctx = 0.1 * torch.randn(1, S, 64 if not v23 else cfg.inner_dim, generator=gen)
    actx = 0.1 * torch.randn(1, S, 64 if not v23 else cfg.audio_inner_dim, generator=gen)
    vpos = loop.video_positions(1, f, h, wd, 24.0)
    apos = dit_av.audio_positions(1, Ta)
    sigma = 0.725
    vmask = (torch.rand(1, f * h * wd, 1, generator=gen) > 0.2).float()
    cases = {}
    for tsk, vts, ats in (("scalar", torch.tensor([sigma]), torch.tensor([sigma])),
                          ("pertoken", vmask * sigma, torch.ones(1, Ta, 1) * sigma)):
        video = dict(latent=vlat, context=vctx, timesteps=vts, sigma=torch.tensor([sigma]), positions=vpos)
        audio = dict(latent=alat, context=actx, timesteps=ats, sigma=torch.tensor([sigma]), positions=apos)
        cases[tsk] = (video, audio)
    return cfg, w, cases


@pytest.mark.parametrize("v23", [False, True])
def test_av_dit_matches_reference(v23):
    """AudioVideo DiT (19B-style blocks and the V2.3 variant) against the reference's LTXModel / X0Model."""
    from oracle import dit_av
    z = g("dit_av_tiny.npz")
    tag = "v23" if v23 else "v1"
    cfg, w, cases = av_tiny_case(v23)
    with torch.no_grad():
        for tsk, (video, audio) in cases.items():
            vv, av = dit_av.av_velocity_model(video, audio, w, cfg)
            close(vv, z[f"{tag}_{tsk}_video_velocity"], rtol=2e-4, atol=2e-5)
            close(av, z[f"{tag}_{tsk}_audio_velocity"], rtol=2e-4, atol=2e-5)
            vx0, ax0 = dit_av.av_x0_model(video, audio, w, cfg)
            close(vx0, z[f"{tag}_{tsk}_video_x0"], rtol=2e-4, atol=2e-5)
            close(ax0, z[f"{tag}_{tsk}_audio_x0"], rtol=2e-4, atol=2e-5)


def av_masked_case(v23: bool):
    """The masked case of pin_dit_av: a different boolean context mask per modality, the masked keys' context rows x 40."""
    cfg, w, cases = av_tiny_case(v23)
    video, audio = dict(cases["scalar"][0]), dict(cases["scalar"][1])
    S = video["context"].shape[1]
    vm = torch.ones(1, S, dtype=torch.int32)
    vm[0, 10:] = 0
    vm[0, 2] = 0
    am = torch.ones(1, S, dtype=torch.int32)
    am[0, :5] = 0
    video["context"] = video["context"].clone()
    audio["context"] = audio["context"].clone()
    video["context"][0, vm[0] == 0] *= 40.0
    audio["context"][0, am[0] == 0] *= 40.0
    return cfg, w, video, audio, vm, am


@pytest.mark.parametrize("v23", [False, True])
def test_av_masked_text_cross_attention_matches_reference(v23):
    """Each modality's Modality.context_mask reaches ITS text cross-attention (transformer.py:523, 551): the oracle with both masks against the
    reference's X0Model output, and without them against the recorded control (which differs: the masked keys' rows are 40x larger)."""
    from oracle import dit_av
    z = g("dit_av_tiny.npz")
    tag = "v23" if v23 else "v1"
    cfg, w, video, audio, vm, am = av_masked_case(v23)
    with torch.no_grad():
        vx0, ax0 = dit_av.av_x0_model(dict(video, context_mask=vm), dict(audio, context_mask=am), w, cfg)
        close(vx0, z[f"{tag}_masked_video_x0"], rtol=2e-4, atol=2e-5)
        close(ax0, z[f"{tag}_masked_audio_x0"], rtol=2e-4, atol=2e-5)
        vx0, ax0 = dit_av.av_x0_model(video, audio, w, cfg)
        close(vx0, z[f"{tag}_masked_control_video_x0"], rtol=2e-4, atol=2e-5)
        close(ax0, z[f"{tag}_masked_control_audio_x0"], rtol=2e-4, atol=2e-5)
    assert float(np.abs(z[f"{tag}_masked_video_x0"] - z[f"{tag}_masked_control_video_x0"]).max()) > 5e-2
    assert float(np.abs(z[f"{tag}_masked_audio_x0"] - z[f"{tag}_masked_control_audio_x0"]).max()) > 1e-2


def av_videoonly_case(v23: bool):
    """The video-only-inference case of pin_dit_av: the AudioVideo model called without audio, per-token timesteps of an image-
    conditioned state (token 0 clean, first latent frame at strength 0.95), Modality.sigma set."""
    cfg, w, cases = av_tiny_case(v23)
    video = dict(cases["scalar"][0])
    f, h, wd, sigma = 3, 4, 4, 0.725
    cmask = torch.ones(1, f * h * wd, 1)
    cmask[:, :h * wd] = 0.05
    cmask[:, 0] = 0.0
    video["timesteps"] = cmask * sigma
    return cfg, w, video


@pytest.mark.parametrize("v23", [False, True])
def test_video_only_inference_matches_reference(v23):
    """oracle.dit_av.video_only_x0_model against the reference's X0Model(LTXModel(AudioVideo))(video, None) (model.py:829-840),
    with timesteps[0] = 0 != sigma: pins that the prompt AdaLN of the V2.3 blocks follows Modality.sigma (model.py:151-158)."""
    from oracle import dit_av
    z = g("dit_av_tiny.npz")
    cfg, w, video = av_videoonly_case(v23)
    with torch.no_grad():
        x0 = dit_av.video_only_x0_model(video, w, cfg)
        close(x0, z[f"{'v23' if v23 else 'v1'}_videoonly_x0"], rtol=2e-4, atol=2e-5)
        if v23:     # negative control: sigma taken from timesteps[0] (= 0) is NOT what the reference computes
            wrong = dict(video, sigma=video["timesteps"].reshape(-1)[:1])
            assert (dit_av.video_only_x0_model(wrong, w, cfg) - torch.from_numpy(z["v23_videoonly_x0"])).abs().max() > 1e-3


def test_upscaler_matches_reference():
    """Spatial x2 upscaler oracle against the reference's SpatialUpscaler (tiny: 64 -> 64 channels, 2+2 blocks)."""
    from oracle import upscaler
    z = g("upscaler_tiny.npz")
    w = upscaler.make_upscaler_weights(64, 64, 2, seed=41)
    x = torch.randn(1, 64, 3, 5, 6, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        close(upscaler.spatial_upscaler(x, w, num_blocks=2), z["upscaled"], rtol=2e-4, atol=2e-5)


def test_audio_host_logic_matches_reference():
    """Product host-side audio helpers (AudioLatentShape, AudioPatchifier, AudioLatentTools, the length-invariant
    audio noise normalisation of DistilledPipeline) and the oracle's audio_positions against the reference."""
    from oracle import dit_av
    from ltx_2_mlx_amd.components import AudioPatchifier
    from ltx_2_mlx_amd.conditioning import AudioLatentTools
    from ltx_2_mlx_amd.pipelines import DistilledPipeline
    from ltx_2_mlx_amd.types import AudioLatentShape, VideoPixelShape
    z = g("loop.npz")
    shp = AudioLatentShape.from_video_pixel_shape(VideoPixelShape(batch=1, frames=65, height=512, width=768, fps=24.0))
    assert list(shp.to_tuple()) == list(z["audio_shape_65f_24fps"])
    st = AudioLatentTools(patchifier=AudioPatchifier(patch_size=1), target_shape=AudioLatentShape(1, 8, 11, 16)).create_initial_state()
    close(st.positions, z["audio_positions_11"], rtol=1e-6, atol=1e-7)
    close(dit_av.audio_positions(1, 11), z["audio_positions_11"], rtol=1e-6, atol=1e-7)
    assert st.latent.shape == (1, 11, 128) and st.denoise_mask.shape == (1, 11, 1)
    gen = torch.Generator().manual_seed(5)
    # advance the stream exactly as pin_loop does before its audio draws
    torch.randn(1, 24, 128, generator=gen)
    torch.randn(1, 24, 128, generator=gen)
    torch.rand(1, 24, 1, generator=gen)
    torch.randn(1, 128, 3, 4, 5, generator=gen)
    alat = torch.randn(1, 8, 11, 16, generator=gen)
    close(AudioPatchifier(patch_size=1).patchify(alat), z["audio_patchify"], rtol=0, atol=0)
    tools = AudioLatentTools(patchifier=AudioPatchifier(patch_size=1), target_shape=AudioLatentShape(1, 8, 11, 16))
    assert torch.equal(AudioPatchifier(patch_size=1).unpatchify(AudioPatchifier(patch_size=1).patchify(alat), tools.target_shape), alat)
    anoise = torch.randn(1, 37, 128, generator=gen) * 1.7 + 0.3
    close(DistilledPipeline._channelwise_normalize_audio(anoise), z["audio_channelwise_normalize"], rtol=1e-4, atol=1e-5)


def test_vae_encoder_matches_reference():
    """VAE encoder oracle (full-size widths, the reference hard-wires them) against the reference's
    SimpleVideoEncoder on one image and on a 9-frame clip."""
    from oracle import vae_encoder
    z = g("vae_encoder.npz")
    w = vae_encoder.make_encoder_weights(seed=51)
    gen = torch.Generator().manual_seed(52)
    img = torch.rand(1, 3, 1, 64, 64, generator=gen) * 2 - 1
    clip = torch.rand(1, 3, 9, 64, 96, generator=gen) * 2 - 1
    with torch.no_grad():
        close(vae_encoder.encoder_forward(img, w), z["image_latent"], rtol=5e-4, atol=5e-5)
        close(vae_encoder.encoder_forward(clip, w), z["clip_latent"], rtol=5e-4, atol=5e-5)
    with pytest.raises(ValueError, match="1 \\+ 8\\*k frames"):
        vae_encoder.encoder_forward(torch.zeros(1, 3, 4, 64, 64), w)


def test_latent_index_conditioning_matches_reference():
    """Product VideoConditionByLatentIndex.apply_to against the reference (strength 0.8 at latent frame 0)."""
    from ltx_2_mlx_amd.components import VideoLatentPatchifier
    from ltx_2_mlx_amd.conditioning import ConditioningError, VideoConditionByLatentIndex, VideoLatentTools
    from ltx_2_mlx_amd.types import VideoLatentShape
    z = g("vae_encoder.npz")
    tools = VideoLatentTools(VideoLatentPatchifier(1), VideoLatentShape(1, 128, 3, 2, 2), fps=24.0)
    st = tools.create_initial_state()
    st2 = VideoConditionByLatentIndex(latent=torch.from_numpy(z["image_latent"]), strength=0.8, latent_idx=0).apply_to(st, tools)
    close(st2.latent, z["cond_latent"], rtol=0, atol=0)
    close(st2.denoise_mask, z["cond_mask"], rtol=1e-6, atol=1e-7)
    close(st2.clean_latent, z["cond_clean"], rtol=0, atol=0)
    with pytest.raises(ConditioningError):
        VideoConditionByLatentIndex(latent=torch.zeros(1, 128, 1, 3, 2), strength=1.0, latent_idx=0).apply_to(st, tools)
    with pytest.raises(ValueError, match="exceed latent sequence length"):
        VideoConditionByLatentIndex(latent=torch.zeros(1, 128, 2, 2, 2), strength=1.0, latent_idx=2).apply_to(st, tools)


def test_text_connector_and_feature_extractors_match_reference():
    """Embeddings1DConnector (registers, INTERLEAVED RoPE with the fp32 and the float64 grid) and both Gemma feature
    extractors, from the seeds of tools/pin_oracle_against_reference.py::pin_text_connector."""
    from oracle import text_connector as tc
    z = g("text_connector.npz")
    for tag, dbl in (("f32", False), ("f64", True)):
        cfg = tc.ConnectorConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, num_learnable_registers=16,
                                 double_precision_rope=dbl)
        w = tc.make_connector_weights(cfg, seed=61)
        x = torch.randn(1, 40, cfg.inner_dim, generator=torch.Generator().manual_seed(62))
        y, mask = tc.encode_projected(x, torch.ones(1, 40), w, cfg)
        assert y.shape == (1, 1024, 256) and int(mask.sum()) == 1024
        close(y[:, :56], z[f"connector_{tag}_head"], rtol=2e-4, atol=2e-5)
        close(y[:, 992:], z[f"connector_{tag}_tail"], rtol=2e-4, atol=2e-5)
        close(torch.stack([y.mean(), y.std(unbiased=False), y.abs().max()]), z[f"connector_{tag}_stats"], rtol=1e-4, atol=1e-6)
    gen = torch.Generator().manual_seed(63)
    hs = [torch.randn(2, 12, 32, generator=gen) * (1 + 0.3 * i) + 0.1 * i for i in range(5)]
    am = torch.ones(2, 12)
    am[1, :5] = 0
    w1 = 0.05 * torch.randn(32, 160, generator=gen)
    close(tc.feature_extractor_v1(hs, am, {"aggregate_embed.weight": w1}, "left"), z["fe_v1_left"], rtol=1e-5, atol=1e-6)
    am_r = torch.ones(2, 12)
    am_r[1, 7:] = 0
    close(tc.feature_extractor_v1(hs, am_r, {"aggregate_embed.weight": w1}, "right"), z["fe_v1_right"], rtol=1e-5, atol=1e-6)
    wv, bv = 0.05 * torch.randn(48, 160, generator=gen), 0.1 * torch.randn(48, generator=gen)
    wa, ba = 0.05 * torch.randn(24, 160, generator=gen), 0.1 * torch.randn(24, generator=gen)
    v, a = tc.feature_extractor_v2(hs, am, {"video_aggregate_embed.weight": wv, "video_aggregate_embed.bias": bv,
                                            "audio_aggregate_embed.weight": wa, "audio_aggregate_embed.bias": ba})
    close(v, z["fe_v2_video"], rtol=1e-5, atol=1e-6)
    close(a, z["fe_v2_audio"], rtol=1e-5, atol=1e-6)
