"""Model-level parity on MI355X: the HIP path (through the C-ABI engines, driven through the
reference-shaped Python API) against the fp32 CPU oracle, on identical seeded latents /
text embeddings / weights.

Tolerance (bf16 operands, fp32 accumulation and fp32 residual stream, vs an fp32 oracle):
  * DiT x0 / velocity: relative L2 <= 2e-2 per call, Pearson r >= 0.999
    (the reference's own bar vs upstream PyTorch is Pearson r >= 0.95, tests/test_parity.py:38)
  * 8-step sampled latent: relative L2 <= 3e-2, r >= 0.999
  * VAE decode (bf16 activations through ~25 convs): relative L2 <= 4e-2, r >= 0.999;
    uint8 frames: mean abs diff <= 2 levels.
"""
import pytest
import torch

from conftest import measure, rel_l2

pytestmark = pytest.mark.gpu


def pearson(a, b):
    a = a.double().flatten() - a.double().mean()
    b = b.double().flatten() - b.double().mean()
    return float((a * b).sum() / (a.norm() * b.norm()))


def make_dit(dev, heads, layers, cap, seed=0):
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import LTXModel
    cfg = dit.DiTConfig(num_attention_heads=heads, attention_head_dim=128, num_layers=layers, caption_channels=cap)
    w = dit.make_dit_weights(cfg, seed)
    # the oracle sees exactly the bf16-rounded linear weights the engine uses
    wq = {k: (v.to(torch.bfloat16).float() if (k.endswith(".weight") and v.dim() == 2) else v) for k, v in w.items()}
    m = LTXModel(num_attention_heads=heads, attention_head_dim=128, num_layers=layers, caption_channels=cap, device=dev)
    m.load_state_dict(w)
    return cfg, wq, m


def inputs(f, h, w, S, cap, seed=1234):
    from oracle import loop
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, f * h * w, 128, generator=g)
    ctx = 0.1 * torch.randn(1, S, cap, generator=g)
    pos = loop.video_positions(1, f, h, w, 24.0)
    return lat, ctx, pos


@pytest.mark.parametrize("grid,S", [((3, 8, 12), 256), ((3, 4, 4), 64), ((2, 5, 7), 100)])
def test_dit_x0_tiny(dev, grid, S):
    """Plumbing config of BASELINE.json (256x384x17 -> 3x8x12 tokens, 2 layers) and ragged shapes."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    lat, ctx, pos = inputs(*grid, S, 128)
    sigma = torch.tensor([0.909375])
    ref = dit.x0_model(lat, ctx, sigma, pos, w, cfg)
    x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert x0.shape == ref.shape and x0.dtype == torch.float32
    assert rel_l2(x0.cpu(), ref) < 0.0025 and pearson(x0.cpu(), ref) > 0.999


def test_dit_per_token_timesteps(dev):
    """Pipeline path: timesteps (B,N,1) = mask*sigma -> per-token AdaLN (pipelines/common.py:193-232)."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    lat, ctx, pos = inputs(3, 4, 6, 64, 128)
    N = lat.shape[1]
    g = torch.Generator().manual_seed(7)
    mask = (torch.rand(1, N, 1, generator=g) > 0.3).float()
    ts = mask * 0.725
    ref = dit.x0_model(lat, ctx, ts, pos, w, cfg)
    x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
    assert rel_l2(x0.cpu(), ref) < 0.002
    # uniform per-token timesteps stay on the per-token path (no host sync to find out that they are equal -- the pipelines
    # pass a 1-element tensor when they KNOW the mask is uniform): same arithmetic through the per-token AdaLN GEMMs
    uni = torch.full((1, N, 1), 0.725)
    a = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=uni.to(dev), positions=pos.to(dev)))
    b = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=torch.tensor([0.725], device=dev), positions=pos.to(dev)))
    assert rel_l2(a.cpu(), b.cpu()) < 0.0008
    assert rel_l2(a.cpu(), dit.x0_model(lat, ctx, uni, pos, w, cfg)) < 0.002


def test_conditioned_loop_through_the_hipgraph(dev):
    """Image-to-video loops (some tokens conditioned: denoise mask < 1): the captured graph forms timesteps = mask * sigma_i on the device and
    blends x0 with the clean latent inside every step (ltx2_dit_graph_capture_cond) -- against the eager loop of X0Model + post_process_latent +
    EulerDiffusionStep calls (reference pipelines/common.py:193-232, distilled.py:214-253): the same kernels, so the same numbers."""
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, EulerDiffusionStep
    from ltx_2_mlx_amd.model.transformer import X0Model
    from ltx_2_mlx_amd.pipelines.common import joint_denoise_loop
    from ltx_2_mlx_amd.types import LatentState
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    lat, ctx, pos = inputs(3, 4, 6, 64, 128)
    N = lat.shape[1]
    g = torch.Generator().manual_seed(21)
    mask = torch.ones(1, N, 1)
    mask[:, :24] = 0.0                       # the first latent frame is the conditioning image
    mask[:, 24:30] = 0.05                    # a partially noised band
    clean = torch.randn(1, N, 128, generator=g)
    outs = []
    for graph in (False, True):
        st = LatentState(latent=lat.clone().to(dev), denoise_mask=mask.to(dev), positions=pos.to(dev), clean_latent=clean.to(dev))
        vs, _ = joint_denoise_loop(X0Model(m), False, st, None, DISTILLED_SIGMA_VALUES, ctx.to(dev), None, EulerDiffusionStep(), use_hip_graph=graph)
        outs.append(vs.latent.float().cpu())
    assert torch.isfinite(outs[1]).all()
    assert rel_l2(outs[1], outs[0]) < 1e-5
    # the conditioned tokens end at their clean values, the free ones moved
    assert rel_l2(outs[1][:, :24], clean[:, :24]) < 1e-5 and rel_l2(outs[1][:, 30:], lat[:, 30:]) > 0.1
    # a mask / clean latent of another length (the audio modality's, say) is refused at capture: the replay would read past it
    lat_d = lat[0].clone().to(dev)
    with torch.cuda.stream(torch.cuda.Stream()):
        with pytest.raises(ValueError, match="tokens x"):
            m.capture_denoise_graph(lat_d, DISTILLED_SIGMA_VALUES, denoise_mask=torch.ones(N - 8, device=dev), clean_latent=clean[0].to(dev).contiguous())
        with pytest.raises(ValueError, match="tokens x"):
            m.capture_denoise_graph(lat_d, DISTILLED_SIGMA_VALUES, denoise_mask=torch.ones(N, device=dev), clean_latent=clean[0, :N - 8].to(dev).contiguous())


def test_dit_full_width_block(dev):
    """Full-width (D=4096, 32 heads, caption 3840) single block at N=288, S=128."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, m = make_dit(dev, heads=32, layers=1, cap=3840, seed=3)
    lat, ctx, pos = inputs(3, 8, 12, 128, 3840)
    sigma = torch.tensor([0.975])
    ref = dit.velocity_model(lat, ctx, sigma, pos, w, cfg)
    v = m(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert rel_l2(v.cpu(), ref) < 0.012 and pearson(v.cpu(), ref) > 0.999


def test_dit_baseline_size_block(dev):
    """BASELINE.json config 2 geometry: one full-width block at the full 768x512x65 token count (N=3456, S=1024,
    D=4096) against the fp32 oracle -- exercises the 224-row ping-pong GEMM grid (16x16 / 16x64 tiles), the
    attention tail tile (3456 = 54 KV tiles) and the text cross-attention at the sizes bench.py measures."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, m = make_dit(dev, heads=32, layers=1, cap=3840, seed=8)
    lat, ctx, pos = inputs(9, 16, 24, 1024, 3840, seed=77)
    sigma = torch.tensor([0.725])
    ref = dit.velocity_model(lat, ctx, sigma, pos, w, cfg)
    v = m(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert v.shape == (1, 3456, 128)
    assert rel_l2(v.cpu(), ref) < 0.012 and pearson(v.cpu(), ref) > 0.999
    # determinism: the video path has no atomics -- a second call is bit-identical
    v2 = m(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert torch.equal(v, v2)


def test_adaln_rows_combined_per_step_is_bit_identical(dev):
    """Round 4: with one timestep per modality the engine adds the timestep embedding to EVERY layer's scale_shift_table in one launch at the top
    of the step (norm kernels read half the vectors, the gated-residual GEMMs take their gate as a table).  Same arithmetic, same rounding order:
    the velocity is bit-identical to the form that hands tables and embeddings to every kernel separately (option adaln_combine = 0)."""
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, m = make_dit(dev, heads=4, layers=3, cap=128, seed=5)
    lat, ctx, pos = inputs(3, 8, 12, 128, 128, seed=6)
    sigma = torch.tensor([0.6])
    mod = Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev))
    a = m(mod)
    m.set_option("adaln_combine", 0)
    b = m(mod)
    m.set_option("adaln_combine", 1)
    c = m(mod)
    assert torch.equal(a, b) and torch.equal(a, c)


def test_denoise_loop_and_graph(dev):
    """8 distilled steps (CLI loop, scripts/generate.py:1797-1979): API-faithful loop, fused C step
    and hipGraph replay all agree with the oracle (and the two fused forms with each other)."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, EulerDiffusionStep
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    f, h, wd = 3, 8, 12
    lat, ctx, pos = inputs(f, h, wd, 256, 128)
    sig = DISTILLED_SIGMA_VALUES
    assert sig == loop.DISTILLED_SIGMA_VALUES
    ref = loop.denoise_loop_cli(loop.unpatchify(lat, f, h, wd), lambda tok, s: dit.x0_model(tok, ctx, torch.tensor([s]), pos, w, cfg), sig)
    ref = loop.patchify(ref)
    # (a) API-faithful: X0Model + EulerDiffusionStep per step
    x = lat.to(dev)
    x0m, stepper = X0Model(m), EulerDiffusionStep()
    C, P = ctx.to(dev), pos.to(dev)
    for i in range(8):
        x0 = x0m(Modality(latent=x, context=C, context_mask=None, timesteps=torch.tensor([sig[i]], device=dev), positions=P))
        x = stepper.step(x, x0, sig, i)
    assert rel_l2(x.cpu(), ref) < 0.002 and pearson(x.cpu(), ref) > 0.999
    # (b) fused step
    y = lat[0].to(dev).contiguous()
    for i in range(8):
        mod = Modality(latent=y[None], context=C, context_mask=None, timesteps=torch.tensor([sig[i]], device=dev), positions=P)
        m.denoise_step_(y, mod, sig[i], sig[i + 1])
    assert rel_l2(y.cpu(), x[0].cpu()) < 1e-5
    # (c) hipGraph replay (twice: replays are re-entrant on fresh latents)
    for _ in range(2):
        z = lat[0].to(dev).contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.capture_denoise_graph(z, sig)
            m.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(z, y)


def make_vae(dev, base=64, layers=2, tcond=True, seed=5):
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder
    blocks = [["res_x", {"num_layers": layers}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": layers}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": layers}], ["compress_all", {"multiplier": 2, "residual": True}],
              ["res_x", {"num_layers": layers}]]
    cfg = vae.VAEConfig(decoder_blocks=blocks, base_channels=base, timestep_conditioning=tcond)
    w = vae.make_vae_weights(cfg, seed)
    wq = {k: (v.to(torch.bfloat16).float() if (v.dim() == 5 or (v.dim() == 2 and "linear" in k)) else v) for k, v in w.items()}
    d = SimpleVideoDecoder(decoder_blocks=blocks, base_channels=base, timestep_conditioning=tcond, device=dev)
    d.load_state_dict(w)
    return cfg, wq, d


@pytest.mark.parametrize("tcond", [True, False])
def test_vae_decoder_forward(dev, tcond):
    from oracle import vae
    cfg, w, d = make_vae(dev, tcond=tcond)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, 128, 3, 4, 5, generator=g)
    noise = torch.randn(1, 128, 3, 4, 5, generator=g)
    ref = vae.decoder_forward(z, w, cfg, timestep=0.05, noise=noise)
    out = d(z.to(dev), timestep=0.05, noise=noise.to(dev))
    assert out.shape == ref.shape == (1, 3, 17, 128, 160)
    assert rel_l2(out.cpu(), ref) < 0.03 and pearson(out.cpu(), ref) > 0.999


def test_vae_decode_latent_chunked(dev):
    """T'=9 -> chunks [0:7] and [5:9], 9-frame cross-fade, 65 frames (simple_decoder.py:708-790)."""
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import decode_latent
    cfg, w, d = make_vae(dev, layers=1)
    g = torch.Generator().manual_seed(12)
    z = torch.randn(1, 128, 9, 2, 3, generator=g)
    noise = torch.randn(1, 128, 9, 2, 3, generator=g)
    assert vae.temporal_chunks(9) == [(0, 7), (5, 9)]
    ref = vae.decode_latent(z, w, cfg, noise=noise)
    out = decode_latent(z.to(dev), d, noise=noise.to(dev))
    assert out.shape == ref.shape == (65, 64, 96, 3) and out.dtype == torch.uint8
    diff = (out.cpu().int() - ref.int()).abs().float()
    assert measure("uint8 mean |diff|", diff.mean()) < 1.2 and pearson(out.cpu().float(), ref.float()) > 0.999


def test_vae_decode_tiled(dev):
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import SpatialTilingConfig, TemporalTilingConfig, TilingConfig, decode_tiled
    cfg, w, d = make_vae(dev, layers=1, tcond=False)
    g = torch.Generator().manual_seed(13)
    z = torch.randn(1, 128, 4, 4, 6, generator=g)
    tc = TilingConfig(SpatialTilingConfig(96, 32), TemporalTilingConfig(16, 8))
    ref = vae.decode_tiled(z, lambda t: vae.decoder_forward(t, w, cfg, timestep=None), spatial=(96, 32), temporal=(16, 8))
    out = next(decode_tiled(z.to(dev), lambda t, timestep=None: d(t, timestep=None), tc))
    assert out.shape == ref.shape
    assert rel_l2(out.cpu(), ref) < 0.025


def test_distilled_pipeline_api(dev):
    """DistilledPipeline.__call__ (stage 1 + decode) against the oracle's pipeline loop with the
    same supplied initial noise (per-token timesteps path, post_process_latent, Euler)."""
    from oracle import dit, loop, vae
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    vcfg, vw, d = make_vae(dev, layers=1)
    conf = DistilledConfig(height=256, width=384, num_frames=17, seed=1)        # stage 1: 128x192 -> 3x4x6 tokens
    f, h, wd = loop.latent_shape_from_pixels(17, 128, 192)
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(1, f * h * wd, 128, generator=g)
    ctx = 0.1 * torch.randn(1, 64, 128, generator=g)
    pos = loop.video_positions(1, f, h, wd, conf.fps)
    mask = torch.ones(1, f * h * wd, 1)
    tok = loop.gaussian_noiser(torch.zeros_like(noise), mask, noise, 1.0)
    ref_tok = loop.denoise_loop_pipeline(tok, mask, torch.zeros_like(noise),
                                         lambda x, ts, s: dit.x0_model(x, ctx, ts, pos, w, cfg), loop.DISTILLED_SIGMA_VALUES)
    ref_lat = loop.unpatchify(ref_tok, f, h, wd)
    pipe = DistilledPipeline(m, None, None)
    lat = pipe(ctx.to(dev), None, conf, initial_noise=noise.to(dev))
    assert rel_l2(lat.cpu(), ref_lat) < 0.0015
    conf_g = DistilledConfig(height=256, width=384, num_frames=17, seed=1, use_hip_graph=True)
    lat_g = pipe(ctx.to(dev), None, conf_g, initial_noise=noise.to(dev))
    assert rel_l2(lat_g.cpu(), lat.cpu()) < 1e-5
    # with a decoder: uint8 frames
    pipe2 = DistilledPipeline(m, None, d)
    d.generator = torch.Generator(device=dev).manual_seed(0)
    frames = pipe2(ctx.to(dev), None, conf, initial_noise=noise.to(dev))
    assert frames.shape == (17, 128, 192, 3) and frames.dtype == torch.uint8


# ------------------------------------------------------------------------------------------ AudioVideo DiT
def make_av(dev, v23, layers=2, heads=4, seed=17):
    from oracle import dit_av
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType
    cfg = dit_av.AVConfig(num_attention_heads=heads, attention_head_dim=128, audio_heads=heads, audio_head_dim=64,
                          num_layers=layers, caption_channels=None if v23 else 64, cross_attention_adaln=v23,
                          apply_gated_attention=v23)
    w = dit_av.make_av_weights(cfg, seed=seed)
    wq = {k: (v.to(torch.bfloat16).float() if (k.endswith(".weight") and v.dim() == 2) else v) for k, v in w.items()}
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=heads, attention_head_dim=128, num_layers=layers,
                 caption_channels=cfg.caption_channels, cross_attention_adaln=v23, apply_gated_attention=v23,
                 audio_attention_heads=heads, device=dev)
    m.load_state_dict(w)
    return cfg, w, wq, m


def test_prompt_setup_is_cached_for_host_resident_context(dev):
    """The per-prompt setup (caption projection, 2 x layers text K/V GEMMs, RoPE tables) must run ONCE per prompt also when the
    caller keeps context / positions on the host: the cache key names the caller's tensors, not the device copies."""
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128, seed=3)
    lat, ctx, pos = inputs(3, 4, 4, 64, 128, seed=4)
    calls = []
    real = m.prepare
    m.prepare = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    outs = []
    for sg in (1.0, 0.725, 0.421875):
        mod = Modality(latent=lat.to(dev), context=ctx, context_mask=None, timesteps=torch.tensor([sg]), positions=pos)    # ctx, pos: CPU
        outs.append(X0Model(m)(mod))
    assert len(calls) == 1
    mod = Modality(latent=lat.to(dev), context=ctx.clone(), context_mask=None, timesteps=torch.tensor([1.0]), positions=pos)
    again = X0Model(m)(mod)
    assert len(calls) == 2 and torch.equal(again, outs[0])          # a NEW context tensor is a new prompt


@pytest.mark.parametrize("v23", [False, True])
def test_video_only_inference_on_an_audiovideo_model(dev, v23):
    """AudioVideo LTXModel called without an audio modality (reference model.py:829-840: an empty audio stream; the blocks then run
    only the video self-attention / text cross-attention / feed-forward, transformer.py:479-483): the video output equals, bit for
    bit, a VideoOnly model holding the video half of the same weights, and the audio output is the reference's empty (1, 0, 128)."""
    from oracle import dit_av
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, Modality, X0Model
    cfg = dit_av.AVConfig(num_attention_heads=4, attention_head_dim=128, audio_heads=4, audio_head_dim=64, num_layers=2,
                          caption_channels=None if v23 else 256, cross_attention_adaln=v23, apply_gated_attention=v23)
    w = dit_av.make_av_weights(cfg, seed=5 + v23)
    kw = dict(num_attention_heads=4, attention_head_dim=128, num_layers=2, caption_channels=cfg.caption_channels,
              cross_attention_adaln=v23, apply_gated_attention=v23, device=dev)
    av = LTXModel(model_type=LTXModelType.AudioVideo, audio_attention_heads=4, **kw)
    av.load_state_dict(w)
    vo = LTXModel(model_type=LTXModelType.VideoOnly, **kw)
    vo.load_state_dict({k: w[k] for k in vo.expected_weight_shapes()})
    g = torch.Generator().manual_seed(9)
    s = torch.tensor([0.725])
    from oracle import loop
    vid = Modality(latent=torch.randn(1, 3 * 4 * 4, 128, generator=g).to(dev), context=(0.1 * torch.randn(1, 64, cfg.caption_channels or cfg.inner_dim, generator=g)).to(dev),
                   context_mask=None, timesteps=s.to(dev), positions=loop.video_positions(1, 3, 4, 4, 24.0).to(dev), sigma=s.to(dev))
    ref = X0Model(vo)(vid)
    x0v, x0a = X0Model(av)(vid, None)
    assert torch.equal(x0v, ref) and x0a.shape == (1, 0, 128)
    vel, aud = av(vid)
    assert torch.equal(vel, vo(vid)) and aud.shape == (1, 0, 128)
    # an audio modality that is present but DISABLED is the same thing (transformer.py:480: run_ax needs audio.enabled; model.py:868-872)
    off = Modality(latent=torch.randn(1, 10, 128, generator=g).to(dev), context=(0.1 * torch.randn(1, 64, cfg.caption_channels or cfg.audio_inner_dim, generator=g)).to(dev),
                   context_mask=None, timesteps=s.to(dev), positions=dit_av.audio_positions(1, 10).to(dev), enabled=False, sigma=s.to(dev))
    vel2, aud2 = av(vid, off)
    assert torch.equal(vel2, vel) and aud2.shape == (1, 0, 128)


def test_video_dit_against_reference_vectors(dev):
    """The VideoOnly HIP path DIRECTLY against the vectors recorded from the reference's own LTXModel / X0Model
    (tests/golden/dit_tiny.npz: 2 layers, 2 x 128 heads, scalar and per-token timesteps) -- not only through the oracle."""
    import numpy as np
    import os
    from oracle import dit, loop
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, wq, m = make_dit(dev, heads=2, layers=2, cap=64, seed=11)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_tiny.npz"))
    f, h, wd, S = 3, 4, 4, 16
    gen = torch.Generator().manual_seed(1234)            # the recording script's draws, in its order (test_oracle_golden.py)
    lat = torch.randn(1, f * h * wd, 128, generator=gen)
    ctx = 0.1 * torch.randn(1, S, 64, generator=gen)
    pos = loop.video_positions(1, f, h, wd, 24.0)
    ts = torch.tensor([0.725])
    tsp = (torch.rand(1, f * h * wd, 1, generator=gen) > 0.3).float() * 0.909375
    for tag, t in (("scalar", ts), ("pertoken", tsp)):
        mod = Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=t.to(dev), positions=pos.to(dev))
        vel = m(mod).cpu()
        x0 = X0Model(m)(mod).cpu()
        rv, rx = torch.from_numpy(z[f"velocity_{tag}"]), torch.from_numpy(z[f"x0_{tag}"])
        assert rel_l2(vel, rv) < 0.012 and pearson(vel, rv) > 0.999, tag
        assert rel_l2(x0, rx) < 0.003 and pearson(x0, rx) > 0.999, tag


def to_modality(d, dev):
    from ltx_2_mlx_amd.model.transformer import Modality
    return Modality(latent=d["latent"].to(dev), context=d["context"].to(dev), context_mask=None,
                    timesteps=d["timesteps"].to(dev), positions=d["positions"].to(dev), sigma=d["sigma"].to(dev))


@pytest.mark.parametrize("v23", [False, True])
def test_av_dit_against_oracle_and_reference_vectors(dev, v23):
    """AudioVideo X0Model (tiny 2-layer, 4+4 heads) on the GPU vs the fp32 oracle AND vs the vectors
    recorded from the reference's own LTXModel (tests/golden/dit_av_tiny.npz), scalar and per-token."""
    import numpy as np
    import os
    from oracle import dit_av
    from test_oracle_golden import av_tiny_case
    from ltx_2_mlx_amd.model.transformer import X0Model
    cfg, w, cases = av_tiny_case(v23)
    _, _, wq, m = make_av(dev, v23, seed=17 + v23)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_av_tiny.npz"))
    tag = "v23" if v23 else "v1"
    for tsk, (video, audio) in cases.items():
        vx0, ax0 = X0Model(m)(to_modality(video, dev), to_modality(audio, dev))
        rv, ra = dit_av.av_x0_model(video, audio, wq, cfg)
        assert rel_l2(vx0.cpu(), rv) < 0.003 and pearson(vx0.cpu(), rv) > 0.999, tsk
        assert rel_l2(ax0.cpu(), ra) < 0.0025 and pearson(ax0.cpu(), ra) > 0.999, tsk
        assert rel_l2(vx0.cpu(), torch.from_numpy(z[f"{tag}_{tsk}_video_x0"])) < 0.004
        assert rel_l2(ax0.cpu(), torch.from_numpy(z[f"{tag}_{tsk}_audio_x0"])) < 0.003
        vv, av = m(to_modality(video, dev), to_modality(audio, dev))
        ov, oa = dit_av.av_velocity_model(video, audio, wq, cfg)
        assert rel_l2(vv.cpu(), ov) < 1e-2 and rel_l2(av.cpu(), oa) < 1e-2
        # round 4: the AdaLN rows (per-block and cross-modal) summed once per step for every layer vs handed to each kernel in two parts
        m.set_option("adaln_combine", 0)
        vv0, av0 = m(to_modality(video, dev), to_modality(audio, dev))
        m.set_option("adaln_combine", 1)
        assert torch.equal(vv, vv0) and torch.equal(av, av0), tsk


@pytest.mark.parametrize("v23", [False, True])
def test_av_denoise_loop_and_graph(dev, v23):
    """Joint audio+video distilled loop (pipelines/distilled.py:198-271): fused steps, hipGraph replay and
    the oracle loop agree; ragged token counts (N=70, Na=37, S=50)."""
    from oracle import dit_av, loop
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, wq, m = make_av(dev, v23, seed=5)
    g = torch.Generator().manual_seed(3)
    f, h, wd, Ta, S = 2, 5, 7, 37, 50
    vlat = torch.randn(1, f * h * wd, 128, generator=g)
    alat = torch.randn(1, Ta, 128, generator=g)
    cdim_v = cfg.caption_channels or cfg.inner_dim
    cdim_a = cfg.caption_channels or cfg.audio_inner_dim
    vctx = 0.1 * torch.randn(1, S, cdim_v, generator=g)
    actx = 0.1 * torch.randn(1, S, cdim_a, generator=g)
    vpos, apos = loop.video_positions(1, f, h, wd, 24.0), dit_av.audio_positions(1, Ta)
    sigmas = loop.DISTILLED_SIGMA_VALUES[4:]
    # oracle loop
    rv, ra = vlat.clone(), alat.clone()
    for i in range(len(sigmas) - 1):
        s = torch.tensor([sigmas[i]])
        vx0, ax0 = dit_av.av_x0_model(dict(latent=rv, context=vctx, timesteps=s, sigma=s, positions=vpos),
                                      dict(latent=ra, context=actx, timesteps=s, sigma=s, positions=apos), wq, cfg)
        rv, ra = loop.euler_step(rv, vx0, sigmas[i], sigmas[i + 1]), loop.euler_step(ra, ax0, sigmas[i], sigmas[i + 1])
    # fused eager steps
    lv, la = vlat[0].to(dev).contiguous(), alat[0].to(dev).contiguous()
    vc, ac, vp, ap = vctx.to(dev), actx.to(dev), vpos.to(dev), apos.to(dev)
    for i in range(len(sigmas) - 1):
        s = torch.tensor([sigmas[i]], device=dev)
        mv = Modality(latent=lv[None], context=vc, context_mask=None, timesteps=s, positions=vp, sigma=s)
        ma = Modality(latent=la[None], context=ac, context_mask=None, timesteps=s, positions=ap, sigma=s)
        m.denoise_step_(lv, mv, sigmas[i], sigmas[i + 1], audio_latent=la, audio=ma)
    assert rel_l2(lv.cpu(), rv[0]) < 2.5e-3 and rel_l2(la.cpu(), ra[0]) < 2.5e-3
    # hipGraph replay
    gv, ga = vlat[0].to(dev).contiguous(), alat[0].to(dev).contiguous()
    m.prepare(vc, vp, audio_context=ac, audio_positions=ap)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(gv, sigmas, audio_latent=ga)
        m.replay_denoise_graph()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert rel_l2(gv, lv) < 1e-5 and rel_l2(ga, la) < 1e-5


def test_av_text_kv_ahead_is_bit_identical(dev):
    """V2.3 AudioVideo: the video stream's sigma-modulated text K / V projected on the side stream at the top of the layer (option text_kv_ahead = 1,
    the default) against the inline schedule (0): the same kernels on the same operands, only their stream differs -- bit-identical velocities, eager and
    through the captured graph."""
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
    from ltx_2_mlx_amd.model.transformer import Modality
    from oracle import dit_av, loop
    cfg, w, wq, m = make_av(dev, True, seed=15)
    g = torch.Generator().manual_seed(13)
    f, h, wd, Ta, S = 2, 6, 8, 21, 40
    vlat, alat = torch.randn(1, f * h * wd, 128, generator=g), torch.randn(1, Ta, 128, generator=g)
    vctx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.inner_dim, generator=g)
    actx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.audio_inner_dim, generator=g)
    vpos, apos = loop.video_positions(1, f, h, wd, 24.0), dit_av.audio_positions(1, Ta)
    s = torch.tensor([0.725], device=dev)
    mv = Modality(latent=vlat.to(dev), context=vctx.to(dev), context_mask=None, timesteps=s, positions=vpos.to(dev), sigma=s)
    ma = Modality(latent=alat.to(dev), context=actx.to(dev), context_mask=None, timesteps=s, positions=apos.to(dev), sigma=s)
    outs = {}
    for opt in (1, 0, 1):
        m.set_option("text_kv_ahead", opt)
        v, a = m(mv, ma)
        lv, la = vlat[0].to(dev).contiguous(), alat[0].to(dev).contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.capture_denoise_graph(lv, DISTILLED_SIGMA_VALUES[5:], audio_latent=la)
            m.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        outs.setdefault(opt, []).append((v.clone(), a.clone(), lv.clone(), la.clone()))
    ref = outs[0][0]
    for got in outs[1]:
        for x, y in zip(got, ref):
            assert torch.equal(x, y)


def test_fp8_checkpoint_loader(dev, tmp_path):
    """BASELINE config 3 plumbing: an fp8 (e4m3fn + weight_scale) safetensors checkpoint with the reference's
    key scheme goes through load_transformer_weights(use_fp8=True) (GPU dequantisation) and the model matches
    the oracle run on the dequantised weights; non-DiT and audio keys are skipped as the reference does."""
    from safetensors.torch import save_file
    from oracle import dit
    from ltx_2_mlx_amd.loader import is_fp8_checkpoint, load_transformer_weights
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model
    cfg = dit.DiTConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=128)
    w = dit.make_dit_weights(cfg, seed=4)
    ck, wq = {}, {}
    for k, v in w.items():
        full = "model.diffusion_model." + k
        if k.endswith(".weight") and v.dim() == 2 and "transformer_blocks" in k:
            scale = float(v.abs().max() / 448.0)
            q8 = (v / scale).to(torch.float8_e4m3fn)
            ck[full] = q8
            ck[full + "_scale"] = torch.tensor(scale)
            ck[full.replace(".weight", ".input_scale")] = torch.tensor(1.0)          # ignored (fp8_loader.py:87-97)
            wq[k] = (q8.float() * scale).to(torch.bfloat16).float()
        else:
            ck[full] = v.to(torch.bfloat16) if (k.endswith(".weight") and v.dim() == 2) else v
            wq[k] = ck[full].float()
    ck["vae.decoder.conv_in.conv.weight"] = torch.zeros(4)
    ck["model.diffusion_model.audio_patchify_proj.weight"] = torch.zeros(4, 4)
    ck["model.diffusion_model.video_embeddings_connector.x.weight"] = torch.zeros(4)
    path = str(tmp_path / "tiny_fp8.safetensors")
    save_file(ck, path)
    assert is_fp8_checkpoint(path)
    m = LTXModel(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=128, device=dev)
    load_transformer_weights(m, path, strict=True, use_fp8=True)
    lat, ctx, pos = inputs(3, 4, 4, 64, 128)
    sigma = torch.tensor([0.725])
    ref = dit.x0_model(lat, ctx, sigma, pos, wq, cfg)
    x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert rel_l2(x0.cpu(), ref) < 0.002 and pearson(x0.cpu(), ref) > 0.999


def test_spatial_upscaler(dev):
    """BASELINE config 5 building block: SpatialUpscaler (tiny 64-channel, 2+2 blocks) and the un_normalize /
    normalize bracket on the GPU vs the fp32 oracle and vs the vector recorded from the reference."""
    import numpy as np
    import os
    from oracle import upscaler as oup
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler, upscale_latent
    w = oup.make_upscaler_weights(64, 64, 2, seed=41)
    wq = {k: (v.to(torch.bfloat16).float() if v.dim() >= 4 else v) for k, v in w.items()}
    up = SpatialUpscaler(in_channels=64, mid_channels=64, num_blocks_per_stage=2, device=dev)
    up.load_state_dict(w)
    x = torch.randn(1, 64, 3, 5, 6, generator=torch.Generator().manual_seed(77))
    out = up(x.to(dev))
    assert out.shape == (1, 64, 3, 10, 12) and out.dtype == torch.float32
    ref = oup.spatial_upscaler(x, wq, num_blocks=2)
    assert rel_l2(out.cpu(), ref) < 4e-2 and pearson(out.cpu(), ref) > 0.999
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "upscaler_tiny.npz"))
    assert rel_l2(out.cpu(), torch.from_numpy(z["upscaled"])) < 0.04
    g = torch.Generator().manual_seed(5)
    mean, std = torch.randn(64, generator=g), 0.5 + torch.rand(64, generator=g)
    out2 = upscale_latent(x.to(dev), up, mean, std)
    ref2 = oup.upscale_latent(x, wq, mean, std, num_blocks=2)
    assert rel_l2(out2.cpu(), ref2) < 0.02


def test_two_stage_pipeline_with_upscaler(dev):
    """BASELINE config 5 plumbing at toy size: stage 1 (8 steps, half res) -> un_normalize/upscale x2/normalize ->
    re-noise to sigma 0.909375 -> stage 2 (3 steps, full res); eager per-step API and hipGraph replay agree."""
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128)
    vcfg, vw, d = make_vae(dev, layers=1)
    up = SpatialUpscaler(in_channels=128, mid_channels=64, num_blocks_per_stage=1, device=dev)
    up.init_random_weights(seed=3)
    g = torch.Generator().manual_seed(2)
    ctx = 0.1 * torch.randn(1, 64, 128, generator=g)
    noise = torch.randn(1, 3 * 4 * 6, 128, generator=g)
    pipe = DistilledPipeline(m, None, None, spatial_upscaler=up)
    with pytest.raises(ValueError, match="per_channel_statistics"):
        pipe(ctx.to(dev), None, DistilledConfig(height=256, width=384, num_frames=17, seed=1), initial_noise=noise.to(dev))
    pipe = DistilledPipeline(m, None, d, spatial_upscaler=up)
    outs = []
    for graph in (False, True):
        conf = DistilledConfig(height=256, width=384, num_frames=17, seed=1, use_hip_graph=graph)
        pipe.video_decoder = None                       # latent out
        pipe.video_encoder = d                          # statistics provider (ships with the VAE weights)
        lat = pipe(ctx.to(dev), None, conf, initial_noise=noise.to(dev))
        assert lat.shape == (1, 128, 3, 8, 12) and bool(torch.isfinite(lat).all())
        outs.append(lat)
    assert rel_l2(outs[1], outs[0]) < 1e-05


@pytest.mark.parametrize("v23", [False, True])
def test_av_distilled_pipeline(dev, v23):
    """BASELINE config 4 plumbing at toy size: DistilledPipeline on an AudioVideo transformer -- joint audio+video
    stage 1, upscale, joint stage 2 with the re-noised stage-1 audio latent -- returns (video latent, audio latent);
    the per-step API and the hipGraph replay agree."""
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    cfg, w, wq, m = make_av(dev, v23, seed=9)
    vcfg, vw, d = make_vae(dev, layers=1)
    up = SpatialUpscaler(in_channels=128, mid_channels=64, num_blocks_per_stage=1, device=dev)
    up.init_random_weights(seed=3)
    g = torch.Generator().manual_seed(2)
    cv, ca = (cfg.caption_channels or cfg.inner_dim), (cfg.caption_channels or cfg.audio_inner_dim)
    vctx, actx = 0.1 * torch.randn(1, 40, cv, generator=g), 0.1 * torch.randn(1, 40, ca, generator=g)
    pipe = DistilledPipeline(m, d, None, spatial_upscaler=up)          # decoder object only provides the latent statistics
    outs = []
    for graph in (False, True):
        conf = DistilledConfig(height=256, width=384, num_frames=17, seed=1, fps=24.0, audio_enabled=True, use_hip_graph=graph)
        video, audio = pipe(vctx.to(dev), None, conf, audio_encoding=actx.to(dev))
        assert video.shape == (1, 128, 3, 8, 12) and audio.shape == (1, 8, 18, 16)        # 17 frames / 24 fps * 25 latents/s
        assert bool(torch.isfinite(video).all()) and bool(torch.isfinite(audio).all())
        outs.append((video, audio))
    assert rel_l2(outs[1][0], outs[0][0]) < 1e-4 and rel_l2(outs[1][1], outs[0][1]) < 1e-4
    with pytest.raises(ValueError, match="audio_encoding"):
        pipe(vctx.to(dev), None, DistilledConfig(height=256, width=384, num_frames=17, audio_enabled=True))


def test_vae_encoder_and_image_conditioning(dev, tmp_path):
    """Scope row f4: SimpleVideoEncoder on the GPU vs the fp32 oracle and the reference vectors (one image, one
    9-frame clip), then image-to-video through DistilledPipeline: with strength 1.0 the conditioned latent frame
    leaves the loop exactly as encoded; a PNG on disk goes through load_image_tensor."""
    import numpy as np
    import os
    from oracle import vae_encoder as oenc
    from ltx_2_mlx_amd.model.video_vae_encoder import SimpleVideoEncoder
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline, ImageCondition, load_image_tensor
    w = oenc.make_encoder_weights(seed=51)
    wq = {k: (v.to(torch.bfloat16).float() if v.dim() == 5 else v) for k, v in w.items()}
    enc = SimpleVideoEncoder(device=dev)
    enc.load_state_dict(w)
    gen = torch.Generator().manual_seed(52)
    img = torch.rand(1, 3, 1, 64, 64, generator=gen) * 2 - 1
    clip = torch.rand(1, 3, 9, 64, 96, generator=gen) * 2 - 1
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_encoder.npz"))
    for name, x in (("image_latent", img), ("clip_latent", clip)):
        out = enc(x.to(dev))
        ref = oenc.encoder_forward(x, wq)
        assert out.shape == ref.shape and out.dtype == torch.float32
        assert rel_l2(out.cpu(), ref) < 0.04 and pearson(out.cpu(), ref) > 0.998, name
        assert rel_l2(out.cpu(), torch.from_numpy(z[name])) < 0.05, name
    # image-to-video: 256x384x17 -> stage-1 128x192 image, latent 3x4x6
    cfg, wd, m = make_dit(dev, heads=2, layers=2, cap=128)
    g = torch.Generator().manual_seed(2)
    ctx = 0.1 * torch.randn(1, 64, 128, generator=g)
    image = torch.rand(1, 3, 1, 128, 192, generator=g) * 2 - 1
    pipe = DistilledPipeline(m, enc, None)
    conf = DistilledConfig(height=256, width=384, num_frames=17, seed=1, use_hip_graph=True)     # falls back to per-step (non-uniform mask)
    lat = pipe(ctx.to(dev), None, conf, images=[ImageCondition(None, 0, 1.0, image=image)])
    enc_lat = enc(image.to(dev))
    assert lat.shape == (1, 128, 3, 4, 6)
    assert rel_l2(lat[:, :, 0], enc_lat[:, :, 0]) < 1e-5                  # frame 0 kept clean
    assert float((lat[:, :, 1:] - 0).abs().mean()) > 0.05 and bool(torch.isfinite(lat).all())
    lat2 = pipe(ctx.to(dev), None, conf, images=[ImageCondition(None, 0, 0.5, image=image)])
    assert rel_l2(lat2[:, :, 0], enc_lat[:, :, 0]) > 1e-3                 # partially denoised
    from PIL import Image
    arr = (np.random.RandomState(0).rand(100, 140, 3) * 255).astype(np.uint8)
    path = str(tmp_path / "cond.png")
    Image.fromarray(arr).save(path)
    t = load_image_tensor(path, 128, 192)
    assert t.shape == (1, 3, 1, 128, 192) and float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    lat3 = pipe(ctx.to(dev), None, conf, images=[ImageCondition(path, 0, 1.0)])
    assert lat3.shape == (1, 128, 3, 4, 6) and bool(torch.isfinite(lat3).all())
    with pytest.raises(FileNotFoundError):
        pipe(ctx.to(dev), None, conf, images=[ImageCondition(str(tmp_path / "missing.png"), 0, 1.0)])


def test_lora_fusion_at_load(dev, tmp_path):
    """LoRA adapters (lora_A/lora_B and lora_down/lora_up naming, two adapters with different strengths) are fused
    on the GPU at load: W + sum s_i * B_i @ A_i, and the model matches the oracle run on the fused weights."""
    from safetensors.torch import save_file
    from oracle import dit
    from ltx_2_mlx_amd.loader import LoRAConfig, load_transformer_weights
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model
    cfg = dit.DiTConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=128)
    w = dit.make_dit_weights(cfg, seed=6)
    save_file({"model.diffusion_model." + k: (v.to(torch.bfloat16) if v.dim() == 2 else v) for k, v in w.items()}, str(tmp_path / "base.safetensors"))
    g = torch.Generator().manual_seed(1)
    l1, l2, wf = {}, {}, {k: (v.to(torch.bfloat16).float() if v.dim() == 2 else v.clone()) for k, v in w.items()}
    for i, (name, rank) in enumerate((("transformer_blocks.0.attn1.to_q", 16), ("transformer_blocks.1.ff.net.0.proj", 8), ("transformer_blocks.0.attn2.to_v", 32))):
        o, inn = w[name + ".weight"].shape
        a, b = (torch.randn(rank, inn, generator=g) * 0.05).to(torch.bfloat16), (torch.randn(o, rank, generator=g) * 0.05).to(torch.bfloat16)
        if i < 2:
            l1[f"diffusion_model.{name}.lora_A.weight"], l1[f"diffusion_model.{name}.lora_B.weight"] = a, b
            wf[name + ".weight"] += 0.8 * (b.float() @ a.float())
        if i > 0:
            a2, b2 = (torch.randn(4, inn, generator=g) * 0.1).to(torch.bfloat16), (torch.randn(o, 4, generator=g) * 0.1).to(torch.bfloat16)
            l2[f"{name}.lora_down.weight"], l2[f"{name}.lora_up.weight"] = a2, b2
            wf[name + ".weight"] += -0.5 * (b2.float() @ a2.float())
    save_file(l1, str(tmp_path / "l1.safetensors"))
    save_file(l2, str(tmp_path / "l2.safetensors"))
    with pytest.raises(ValueError, match="between -2.0 and 2.0"):
        LoRAConfig("x", 3.0)
    m = LTXModel(num_attention_heads=2, attention_head_dim=128, num_layers=2, caption_channels=128, device=dev)
    load_transformer_weights(m, str(tmp_path / "base.safetensors"), strict=True,
                             lora_configs=[LoRAConfig(str(tmp_path / "l1.safetensors"), 0.8), LoRAConfig(str(tmp_path / "l2.safetensors"), -0.5)])
    wq = {k: (v.to(torch.bfloat16).float() if v.dim() == 2 else v) for k, v in wf.items()}
    lat, ctx, pos = inputs(3, 4, 4, 64, 128)
    sigma = torch.tensor([0.725])
    ref = dit.x0_model(lat, ctx, sigma, pos, wq, cfg)
    base = dit.x0_model(lat, ctx, sigma, pos, {k: (v.to(torch.bfloat16).float() if v.dim() == 2 else v) for k, v in w.items()}, cfg)
    x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    err = rel_l2(x0.cpu(), ref)
    assert err < 2.5e-3 and rel_l2(base, ref) > 5 * err        # the adapters matter and are applied (measured 5.3e-4)


def test_text_connector_and_feature_extractors(dev, tmp_path):
    """Scope row f3: Embeddings1DConnector (registers -> 1024 tokens, INTERLEAVED RoPE run as SPLIT on per-head permuted
    q/k rows, fp32 and float64 frequency grids) vs the fp32 oracle and the reference vectors; both Gemma feature
    extractors (layer-major re-ordered projection weights) vs the oracle; the safetensors key scheme round trip."""
    import numpy as np
    import os
    from oracle import text_connector as tc
    from ltx_2_mlx_amd.model.text_encoder import (Embeddings1DConnector, GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2,
                                                  VideoGemmaTextEncoderModel, load_text_encoder_weights)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_connector.npz"))

    def qw(w):      # the GPU holds matrices in bf16: give the oracle the same operands
        return {k: (v.to(torch.bfloat16).float() if v.dim() == 2 and k != "learnable_registers" else v) for k, v in w.items()}

    for tag, dbl in (("f32", False), ("f64", True)):
        cfg = tc.ConnectorConfig(num_attention_heads=2, attention_head_dim=128, num_layers=2, num_learnable_registers=16,
                                 double_precision_rope=dbl)
        w = tc.make_connector_weights(cfg, seed=61)
        conn = Embeddings1DConnector(attention_head_dim=128, num_attention_heads=2, num_layers=2, num_learnable_registers=16,
                                     double_precision_rope=dbl, device=dev)
        conn.load_state_dict(w)
        x = torch.randn(1, 40, cfg.inner_dim, generator=torch.Generator().manual_seed(62))
        enc = VideoGemmaTextEncoderModel(feature_extractor=object(), embeddings_connector=conn)
        am = torch.ones(1, 40)
        am[:, 30:] = 0
        out = enc.encode_projected(x.to(dev), am.to(dev))
        ref, _ = tc.encode_projected(x, am, qw(w), cfg)
        y = out.video_encoding.cpu()
        assert y.shape == (1, 1024, 256) and y.dtype == torch.float32 and int(out.attention_mask.sum()) == 1024
        assert rel_l2(y, ref) < 0.008 and pearson(y, ref) > 0.999, tag
        assert rel_l2(y[:, :56], torch.from_numpy(z[f"connector_{tag}_head"])) < 0.008, tag
        assert rel_l2(y[:, 992:], torch.from_numpy(z[f"connector_{tag}_tail"])) < 0.008, tag
    # production width: 30 heads x 128 = 3840, one block, a 100-token prompt
    cfg = tc.ConnectorConfig(num_layers=1)
    w = tc.make_connector_weights(cfg, seed=64)
    conn = Embeddings1DConnector(num_layers=1, device=dev)
    conn.load_state_dict(w)
    x = torch.randn(1, 100, 3840, generator=torch.Generator().manual_seed(65))
    y, m = conn(x.to(dev))
    ref = tc.embeddings_connector(x, qw(w), cfg)
    assert y.shape == (1, 1024, 3840) and float(m.abs().max()) == 0.0
    assert rel_l2(y.cpu(), ref) < 0.012 and pearson(y.cpu(), ref) > 0.999

    # feature extractors: D = 64, L = 5 layers, left-padded batch of 2
    gen = torch.Generator().manual_seed(66)
    hs = [torch.randn(2, 12, 64, generator=gen) * (1 + 0.3 * i) + 0.1 * i for i in range(5)]
    am = torch.ones(2, 12)
    am[1, :5] = 0
    w1 = 0.05 * torch.randn(64, 320, generator=gen)
    fe1 = GemmaFeaturesExtractorProjLinear(hidden_dim=64, num_layers=5, device=dev)
    fe1.load_state_dict({"aggregate_embed.weight": w1})
    o1 = fe1.extract_from_hidden_states([h.to(dev) for h in hs], am.to(dev), padding_side="left").cpu()
    r1 = tc.feature_extractor_v1(hs, am, {"aggregate_embed.weight": w1.to(torch.bfloat16).float()}, "left")
    assert o1.shape == (2, 12, 64) and rel_l2(o1, r1) < 0.006
    assert float(o1[1, :5].abs().max()) == 0.0                     # pad rows: no bias in V1
    wv, bv = 0.05 * torch.randn(128, 320, generator=gen), 0.1 * torch.randn(128, generator=gen)
    wa, ba = 0.05 * torch.randn(64, 320, generator=gen), 0.1 * torch.randn(64, generator=gen)
    fe2 = GemmaFeaturesExtractorV2(hidden_dim=64, num_layers=5, video_inner_dim=128, audio_inner_dim=64, device=dev)
    sd2 = {"video_aggregate_embed.weight": wv, "video_aggregate_embed.bias": bv, "audio_aggregate_embed.weight": wa, "audio_aggregate_embed.bias": ba}
    fe2.load_state_dict(sd2)
    v, a = fe2.extract_from_hidden_states([h.to(dev) for h in hs], am.to(dev))
    rv, ra = tc.feature_extractor_v2(hs, am, {k: (t.to(torch.bfloat16).float() if t.dim() == 2 else t) for k, t in sd2.items()})
    assert rel_l2(v.cpu(), rv) < 1e-2 and rel_l2(a.cpu(), ra) < 1e-2
    assert torch.allclose(v[1, :5].cpu(), bv.expand(5, -1), atol=1e-6)      # pad rows carry the bias only

    # checkpoint key scheme (reference encoder.py:441-520) through safetensors
    from safetensors.torch import save_file
    cfg = tc.ConnectorConfig(num_attention_heads=2, attention_head_dim=128, num_layers=1, num_learnable_registers=16)
    w = tc.make_connector_weights(cfg, seed=67)
    sd = {"model.diffusion_model.video_embeddings_connector." + k: t.contiguous() for k, t in w.items()}
    sd["text_embedding_projection.aggregate_embed.weight"] = w1
    sd["model.diffusion_model.caption_projection.linear_1.weight"] = torch.zeros(4, 4)      # belongs to the transformer: ignored
    path = str(tmp_path / "te.safetensors")
    save_file(sd, path)
    enc = VideoGemmaTextEncoderModel(feature_extractor=GemmaFeaturesExtractorProjLinear(hidden_dim=64, num_layers=5, device=dev),
                                     embeddings_connector=Embeddings1DConnector(attention_head_dim=128, num_attention_heads=2, num_layers=1,
                                                                                num_learnable_registers=16, device=dev))
    assert load_text_encoder_weights(enc, path) == len(w) + 1
    with pytest.raises(ValueError):
        enc.embeddings_connector(torch.zeros(1, 8, 100, device=dev))


def test_vae_full_size_decode(dev):
    """BASELINE config 2 geometry for the decoder (base_channels 128, default blocks, latent 128 x 9 x 16 x 24 ->
    65 x 512 x 768, 37.7 TFLOP with the 7/2 chunking).  The fp32 oracle would take minutes on the host cores, so the
    same oracle code runs in fp32 on the GPU (torch ops only) as the reference; plus bit-for-bit repeatability."""
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, decode_latent
    cfg = vae.VAEConfig()
    w = vae.make_vae_weights(cfg, seed=21)
    d = SimpleVideoDecoder(device=dev)
    d.load_state_dict(w)
    g = torch.Generator().manual_seed(22)
    z = torch.randn(1, 128, 9, 16, 24, generator=g)
    nz = torch.randn(1, 128, 9, 16, 24, generator=g)           # the decode noise is an input here (fixed)
    a = decode_latent(z.to(dev), d, noise=nz.to(dev))
    b = decode_latent(z.to(dev), d, noise=nz.to(dev))
    assert a.shape == (65, 512, 768, 3) and a.dtype == torch.uint8
    assert torch.equal(a, b)
    wq = {k: (v.to(torch.bfloat16).float() if (v.dim() == 5 or (v.dim() == 2 and "linear" in k)) else v).to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ref = vae.decode_latent(z.to(dev), wq, cfg, noise=nz.to(dev))
    assert ref.shape == a.shape
    diff = (a.int() - ref.int()).abs().float()
    assert measure("uint8 mean |diff|", diff.mean()) < 1.2 and pearson(a.float().cpu(), ref.float().cpu()) > 0.999


def test_dit_full_size_denoise_loop(dev):
    """BASELINE config 2 geometry for the DiT (D = 4096, 32 heads, caption 3840, N = 3456 video tokens, S = 1024 text
    tokens) with 4 of the 48 layers: the 8-step distilled loop through the hipGraph against the oracle's loop.  The
    fp32 oracle code runs on the GPU here (torch ops only) -- on the host cores one such step takes a minute."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
    cfg, w, m = make_dit(dev, heads=32, layers=4, cap=3840, seed=31)
    f, h, wd = 9, 16, 24
    lat, ctx, pos = inputs(f, h, wd, 1024, 3840, seed=32)
    sig = DISTILLED_SIGMA_VALUES
    wg = {k: v.to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ctx_g, pos_g = ctx.to(dev), pos.to(dev)
        ref = loop.denoise_loop_cli(loop.unpatchify(lat.to(dev), f, h, wd),
                                    lambda tok, s: dit.x0_model(tok, ctx_g, torch.tensor([s]), pos_g, wg, cfg), sig)
        ref = loop.patchify(ref).cpu()
    m.prepare(ctx.to(dev), pos.to(dev))
    z = lat[0].to(dev).contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(z, sig)
        m.replay_denoise_graph()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert z.shape == (3456, 128)
    assert rel_l2(z.cpu(), ref[0]) < 0.008 and pearson(z.cpu(), ref[0]) > 0.999


@pytest.mark.parametrize("v23", [False, True])
def test_av_full_size_step(dev, v23):
    """BASELINE config 4 geometry: AudioVideo DiT at full width (video 32 x 128, audio 32 x 64), N = 3456 video tokens,
    68 audio tokens, S = 1024, two layers (19B-style and the V2.3 variant with 9-row AdaLN / prompt modulation / head
    gates): one joint x0 evaluation against the fp32 oracle executed on the GPU."""
    from oracle import dit_av, loop
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, X0Model
    cfg = dit_av.AVConfig(num_attention_heads=32, attention_head_dim=128, audio_heads=32, audio_head_dim=64, num_layers=2,
                          caption_channels=None if v23 else 3840, cross_attention_adaln=v23, apply_gated_attention=v23)
    w = dit_av.make_av_weights(cfg, seed=71 + v23)
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=32, attention_head_dim=128, num_layers=2,
                 caption_channels=cfg.caption_channels, cross_attention_adaln=v23, apply_gated_attention=v23,
                 audio_attention_heads=32, device=dev)
    m.load_state_dict(w)
    g = torch.Generator().manual_seed(72)
    f, h, wd, Ta, S = 9, 16, 24, 68, 1024
    s = torch.tensor([0.725])
    video = dict(latent=torch.randn(1, f * h * wd, 128, generator=g), context=0.1 * torch.randn(1, S, cfg.caption_channels or cfg.inner_dim, generator=g),
                 timesteps=s, sigma=s, positions=loop.video_positions(1, f, h, wd, 24.0))
    audio = dict(latent=torch.randn(1, Ta, 128, generator=g), context=0.1 * torch.randn(1, S, cfg.caption_channels or cfg.audio_inner_dim, generator=g),
                 timesteps=s, sigma=s, positions=dit_av.audio_positions(1, Ta))
    vx0, ax0 = X0Model(m)(to_modality(video, dev), to_modality(audio, dev))
    wg = {k: (v.to(torch.bfloat16).float() if (k.endswith(".weight") and v.dim() == 2) else v).to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        rv, ra = dit_av.av_x0_model({k: t.to(dev) for k, t in video.items()}, {k: t.to(dev) for k, t in audio.items()}, wg, cfg)
    assert vx0.shape == (1, 3456, 128) and ax0.shape == (1, 68, 128)
    assert rel_l2(vx0.cpu(), rv.cpu()) < 0.01 and pearson(vx0.cpu(), rv.cpu()) > 0.999
    assert rel_l2(ax0.cpu(), ra.cpu()) < 0.006 and pearson(ax0.cpu(), ra.cpu()) > 0.999


def test_upscaler_and_encoder_full_size(dev):
    """BASELINE config 5 building blocks at their real sizes against the fp32 oracle executed on the GPU: the spatial
    upscaler (128 -> 1024 channels, 4 + 4 blocks) on the 768x512x65 stage-1 latent (9 x 16 x 24 -> 9 x 32 x 48), and the VAE
    encoder on one 512 x 768 conditioning image."""
    from oracle import upscaler as oup, vae_encoder as oenc
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.model.video_vae_encoder import SimpleVideoEncoder
    w = oup.make_upscaler_weights(128, 1024, 4, seed=81)
    up = SpatialUpscaler(device=dev)
    up.load_state_dict(w)
    x = torch.randn(1, 128, 9, 16, 24, generator=torch.Generator().manual_seed(82))
    out = up(x.to(dev))
    wg = {k: (v.to(torch.bfloat16).float() if v.dim() >= 4 else v).to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ref = oup.spatial_upscaler(x.to(dev), wg, num_blocks=4)
    assert out.shape == ref.shape == (1, 128, 9, 32, 48)
    assert rel_l2(out.cpu(), ref.cpu()) < 4e-2 and pearson(out.cpu(), ref.cpu()) > 0.999
    del up, wg, ref
    we = oenc.make_encoder_weights(seed=83)
    enc = SimpleVideoEncoder(device=dev)
    enc.load_state_dict(we)
    img = torch.rand(1, 3, 1, 512, 768, generator=torch.Generator().manual_seed(84)) * 2 - 1
    lat = enc(img.to(dev))
    weg = {k: (v.to(torch.bfloat16).float() if v.dim() == 5 else v).to(dev) for k, v in we.items()}
    with torch.device(dev), torch.no_grad():
        rlat = oenc.encoder_forward(img.to(dev), weg)
    assert lat.shape == rlat.shape == (1, 128, 1, 16, 24)
    assert rel_l2(lat.cpu(), rlat.cpu()) < 0.04 and pearson(lat.cpu(), rlat.cpu()) > 0.998


def test_dit_full_size_per_token_timesteps(dev):
    """Image-to-video at the BASELINE geometry: per-token timesteps (B, N, 1) = mask * sigma with the first latent frame
    conditioned (sigma 0), D = 4096, N = 3456, S = 1024, two layers -- the per-token AdaLN path (N x 6D modulation
    rows, norm_mod with a row stride) against the fp32 oracle executed on the GPU."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, m = make_dit(dev, heads=32, layers=2, cap=3840, seed=41)
    lat, ctx, pos = inputs(9, 16, 24, 1024, 3840, seed=42)
    N = lat.shape[1]
    mask = torch.ones(1, N, 1)
    mask[:, :16 * 24] = 0.05                    # conditioned first latent frame (strength 0.95)
    ts = mask * 0.909375
    x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
    wg = {k: v.to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ref = dit.x0_model(lat.to(dev), ctx.to(dev), ts.to(dev), pos.to(dev), wg, cfg).cpu()
    assert x0.shape == (1, 3456, 128)
    assert rel_l2(x0.cpu(), ref) < 0.012 and pearson(x0.cpu(), ref) > 0.999


@pytest.mark.parametrize("family", ["video", "av_v1", "av_v23"])
def test_one_stage_pipeline_against_oracle_loop(dev, family):
    """OneStagePipeline, guidance-free branch (reference pipelines/one_stage.py:731-1011 with cfg_scale = audio_cfg_scale = 1:
    LTX2Scheduler.execute(steps) sigmas, fps-25 positions, Gaussian noise at scale 1, one X0 evaluation per step, Euler) against
    the same sequence written with the oracle's pieces on the same supplied noise; eager per-step API and hipGraph replay agree.
    `av_v23` is BASELINE config 4's path (LTX-2.3: AudioVideo transformer, prompt AdaLN, gated attention, joint audio+video)."""
    from oracle import dit, dit_av, loop
    from ltx_2_mlx_amd.components import AudioPatchifier
    from ltx_2_mlx_amd.pipelines import OneStageCFGConfig, OneStagePipeline
    from ltx_2_mlx_amd.types import AudioLatentShape
    H, W, F, steps, S = 64, 96, 17, 5, 24
    f, h, wd = 3, 2, 3
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(1, f * h * wd, 128, generator=g)
    vpos = loop.video_positions(1, f, h, wd, 25.0)
    sig = loop.ltx2_scheduler(steps)
    kw = dict(height=H, width=W, num_frames=F, seed=1, fps=25.0, num_inference_steps=steps, cfg_scale=1.0, audio_cfg_scale=1.0, rescale_scale=0.0)
    if family == "video":
        cfg, wq, m = make_dit(dev, heads=2, layers=2, cap=64, seed=11)
        ctx = 0.1 * torch.randn(1, S, 64, generator=g)
        rv = noise.clone()
        for i in range(steps):
            s = float(sig[i])
            rv = loop.euler_step(rv, dit.x0_model(rv, ctx, torch.tensor([s]), vpos, wq, cfg), s, float(sig[i + 1]))
        pipe = OneStagePipeline(m, None, None)
        lat, aud = pipe(ctx.to(dev), None, OneStageCFGConfig(**kw), initial_noise=noise.to(dev))
        assert aud is None and rel_l2(lat.cpu(), loop.unpatchify(rv, f, h, wd)) < 0.002
        lat_g, _ = pipe(ctx.to(dev), None, OneStageCFGConfig(use_hip_graph=True, **kw), initial_noise=noise.to(dev))
        assert rel_l2(lat_g.cpu(), lat.cpu()) < 1e-5
        # classifier-free guidance (one_stage.py:224-330): two evaluations per step, CFGStarRescalingGuider for rescale_scale > 0 else CFGGuider
        from ltx_2_mlx_amd.components import CFGGuider, CFGStarRescalingGuider
        nctx = 2.0 * torch.randn(1, S, 64, generator=g)        # (a tiny random DiT hardly listens to its prompt: a loud negative prompt and a
        SC = 10.0                                               #  large scale make the guided trajectory measurably different)
        for rescale, guider in ((0.7, CFGStarRescalingGuider(scale=SC)), (0.0, CFGGuider(scale=SC))):
            rv = noise.clone()
            for i in range(steps):
                s = float(sig[i])
                pos, ngt = dit.x0_model(rv, ctx, torch.tensor([s]), vpos, wq, cfg), dit.x0_model(rv, nctx, torch.tensor([s]), vpos, wq, cfg)
                rv = loop.euler_step(rv, guider.guide(pos, ngt), s, float(sig[i + 1]))
            lat_c, _ = pipe(ctx.to(dev), nctx.to(dev), OneStageCFGConfig(**dict(kw, cfg_scale=SC, rescale_scale=rescale)), initial_noise=noise.to(dev))
            ref_c = loop.unpatchify(rv, f, h, wd)
            assert rel_l2(lat_c.cpu(), ref_c) < 0.02, rescale
            assert rel_l2(ref_c, lat.cpu()) > 2e-2 and rel_l2(lat_c.cpu(), ref_c) < 0.5 * rel_l2(lat_c.cpu(), lat.cpu())      # the guided trajectory, not the plain one
        with pytest.raises(ValueError, match="negative"):
            pipe(ctx.to(dev), None, OneStageCFGConfig(**dict(kw, cfg_scale=3.0)))
        return
    v23 = family == "av_v23"
    cfg, w, wq, m = make_av(dev, v23, seed=23)
    Ta = AudioLatentShape.from_duration(1, F / 25.0).frames                  # 17 frames at 25 fps -> 17 audio latents
    anoise = torch.randn(1, Ta, 128, generator=g)
    vctx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.inner_dim, generator=g)
    actx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.audio_inner_dim, generator=g)
    apos = dit_av.audio_positions(1, Ta)
    rv, ra = noise.clone(), anoise.clone()
    for i in range(steps):
        s = torch.tensor([float(sig[i])])
        vx0, ax0 = dit_av.av_x0_model(dict(latent=rv, context=vctx, timesteps=s, sigma=s, positions=vpos),
                                      dict(latent=ra, context=actx, timesteps=s, sigma=s, positions=apos), wq, cfg)
        rv, ra = loop.euler_step(rv, vx0, float(sig[i]), float(sig[i + 1])), loop.euler_step(ra, ax0, float(sig[i]), float(sig[i + 1]))
    pipe = OneStagePipeline(m, None, None)
    with pytest.raises(ValueError, match="Audio encoding required"):
        pipe(vctx.to(dev), None, OneStageCFGConfig(**kw))
    lat, aud = pipe(vctx.to(dev), None, OneStageCFGConfig(audio_enabled=True, **kw), positive_audio_encoding=actx.to(dev),
                    initial_noise=noise.to(dev), initial_audio_noise=anoise.to(dev))
    assert rel_l2(lat.cpu(), loop.unpatchify(rv, f, h, wd)) < 0.0025
    ref_aud = AudioPatchifier(patch_size=1).unpatchify(ra, AudioLatentShape(1, 8, Ta, 16))
    assert aud.shape == (1, 8, Ta, 16) and rel_l2(aud.cpu(), ref_aud) < 0.0015
    # silent video from an AV checkpoint (audio_enabled=False): the internal audio branch still runs (use_internal_audio_branch),
    # only the audio output is dropped; hipGraph replay of the joint loop agrees with the eager per-step API
    lat_g, aud_g = pipe(vctx.to(dev), None, OneStageCFGConfig(use_hip_graph=True, **kw), positive_audio_encoding=actx.to(dev),
                        initial_noise=noise.to(dev), initial_audio_noise=anoise.to(dev))
    assert aud_g is None and rel_l2(lat_g.cpu(), lat.cpu()) < 1e-5
    # classifier-free guidance on the joint loop (one_stage.py:466-568): one guider per modality, the reference's default scales 3 / 7
    from ltx_2_mlx_amd.components import CFGStarRescalingGuider
    nvctx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.inner_dim, generator=g)
    nactx = 0.1 * torch.randn(1, S, cfg.caption_channels or cfg.audio_inner_dim, generator=g)
    gv, ga = CFGStarRescalingGuider(scale=3.0), CFGStarRescalingGuider(scale=7.0)
    rv, ra = noise.clone(), anoise.clone()
    for i in range(steps):
        s = torch.tensor([float(sig[i])])
        mv, ma = dict(latent=rv, timesteps=s, sigma=s, positions=vpos), dict(latent=ra, timesteps=s, sigma=s, positions=apos)
        pv, pa = dit_av.av_x0_model(dict(mv, context=vctx), dict(ma, context=actx), wq, cfg)
        nv_, na_ = dit_av.av_x0_model(dict(mv, context=nvctx), dict(ma, context=nactx), wq, cfg)
        rv = loop.euler_step(rv, gv.guide(pv, nv_), float(sig[i]), float(sig[i + 1]))
        ra = loop.euler_step(ra, ga.guide(pa, na_), float(sig[i]), float(sig[i + 1]))
    lat_c, aud_c = pipe(vctx.to(dev), nvctx.to(dev), OneStageCFGConfig(audio_enabled=True, **dict(kw, cfg_scale=3.0, audio_cfg_scale=7.0, rescale_scale=0.7)),
                        positive_audio_encoding=actx.to(dev), negative_audio_encoding=nactx.to(dev), initial_noise=noise.to(dev), initial_audio_noise=anoise.to(dev))
    assert rel_l2(lat_c.cpu(), loop.unpatchify(rv, f, h, wd)) < 0.006
    assert rel_l2(aud_c.cpu(), AudioPatchifier(patch_size=1).unpatchify(ra, AudioLatentShape(1, 8, Ta, 16))) < 0.008
    # use_internal_audio_branch=False on an AV model: the video half alone (reference model.py:829-840), no audio encoding needed
    lat_v, _ = pipe(vctx.to(dev), None, OneStageCFGConfig(use_internal_audio_branch=False, **kw), initial_noise=noise.to(dev))
    assert lat_v.shape == lat.shape and torch.isfinite(lat_v).all() and rel_l2(lat_v.cpu(), lat.cpu()) > 1e-3


def test_video_only_v23_conditioned_token0_against_oracle(dev):
    """VideoOnly LTX-2.3 blocks (cross_attention_adaln + gated attention) under image conditioning: per-token timesteps =
    denoise_mask * sigma with token 0 (and the whole first latent frame) CONDITIONED, so timesteps[0] = 0 != Modality.sigma.
    The prompt AdaLN must take Modality.sigma (reference model.py:151-158), not timesteps[0] -- checked against the oracle
    (oracle/dit_av.py restates that rule), not against another engine configuration."""
    from oracle import dit_av, loop
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, Modality, X0Model
    heads = 4
    cfg = dit_av.AVConfig(num_attention_heads=heads, attention_head_dim=128, audio_heads=heads, audio_head_dim=64, num_layers=2,
                          caption_channels=None, cross_attention_adaln=True, apply_gated_attention=True)
    w = dit_av.make_av_weights(cfg, seed=31)
    wq = {k: (v.to(torch.bfloat16).float() if (k.endswith(".weight") and v.dim() == 2) else v) for k, v in w.items()}
    vo = LTXModel(model_type=LTXModelType.VideoOnly, num_attention_heads=heads, attention_head_dim=128, num_layers=2, caption_channels=None,
                  cross_attention_adaln=True, apply_gated_attention=True, device=dev)
    vo.load_state_dict({k: w[k] for k in vo.expected_weight_shapes()})
    g = torch.Generator().manual_seed(32)
    f, h, wd, S, sigma = 3, 4, 5, 40, 0.909375
    n = f * h * wd
    mask = torch.ones(1, n, 1)
    mask[:, :h * wd] = 0.05                                     # first latent frame conditioned at strength 0.95
    mask[:, 0] = 0.0                                            # token 0 fully clean: timesteps[0] == 0
    ts = mask * sigma
    video = dict(latent=torch.randn(1, n, 128, generator=g), context=0.1 * torch.randn(1, S, cfg.inner_dim, generator=g),
                 timesteps=ts, sigma=torch.tensor([sigma]), positions=loop.video_positions(1, f, h, wd, 24.0))
    rv = dit_av.video_only_x0_model(video, wq, cfg)
    mod = Modality(latent=video["latent"].to(dev), context=video["context"].to(dev), context_mask=None, timesteps=ts.to(dev),
                   positions=video["positions"].to(dev), sigma=video["sigma"].to(dev))
    x0 = X0Model(vo)(mod).cpu()
    assert rel_l2(x0, rv) < 0.003 and pearson(x0, rv) > 0.999
    # (that the oracle itself follows Modality.sigma, not timesteps[0], is pinned with the reference's own vector and a negative
    # control in tests/test_oracle_golden.py::test_video_only_inference_matches_reference)


@pytest.mark.parametrize("v23", [False, True])
def test_video_only_inference_against_reference_vectors(dev, v23):
    """AudioVideo LTXModel called with video alone, image-conditioned per-token timesteps (token 0 clean) and Modality.sigma: the HIP
    path DIRECTLY against the vector recorded from the reference's own model (tests/golden/dit_av_tiny.npz, `*_videoonly_x0`)."""
    import numpy as np
    import os
    from test_oracle_golden import av_videoonly_case
    from ltx_2_mlx_amd.model.transformer import X0Model
    cfg, w, video = av_videoonly_case(v23)
    _, _, wq, m = make_av(dev, v23, seed=17 + v23)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_av_tiny.npz"))
    x0, empty = X0Model(m)(to_modality(video, dev), None)
    ref = torch.from_numpy(z[f"{'v23' if v23 else 'v1'}_videoonly_x0"])
    assert empty.shape == (1, 0, 128) and rel_l2(x0.cpu(), ref) < 0.004 and pearson(x0.cpu(), ref) > 0.999


def test_masked_text_cross_attention_against_oracle_and_reference_vectors(dev):
    """Modality.context_mask (boolean / 0-1 key mask, model.py:163-201 -> attention.py:38-70): the HIP path against the vector recorded from
    the reference's own X0Model (tests/golden/dit_tiny.npz `x0_masked`: padded tail + one hole, the masked keys' context rows x 40 so that
    ignoring the mask lands on `x0_masked_control` instead), against the oracle on a ragged shape, and cleared again by a mask-less call."""
    import numpy as np
    import os
    from oracle import dit, loop
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    from test_oracle_golden import masked_case
    cfg, wq, m = make_dit(dev, heads=2, layers=2, cap=64, seed=11)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_tiny.npz"))
    f, h, wd, S = 3, 4, 4, 16
    gen = torch.Generator().manual_seed(1234)
    lat = torch.randn(1, f * h * wd, 128, generator=gen)
    ctx = 0.1 * torch.randn(1, S, 64, generator=gen)
    pos = loop.video_positions(1, f, h, wd, 24.0)
    ts = torch.tensor([0.725])
    cmask, ctx_m = masked_case(ctx)
    x0m = X0Model(m)

    def run(mask):
        return x0m(Modality(latent=lat.to(dev), context=ctx_m.to(dev), context_mask=None if mask is None else mask.to(dev),
                            timesteps=ts.to(dev), positions=pos.to(dev))).cpu()
    rm, rc = torch.from_numpy(z["x0_masked"]), torch.from_numpy(z["x0_masked_control"])
    for mk in (cmask, cmask.bool(), cmask.to(torch.int64)):
        got = run(mk)
        assert rel_l2(got, rm) < 0.003 and pearson(got, rm) > 0.999
        assert rel_l2(got, rm) < 0.5 * rel_l2(got, rc)            # ... and it is the masked vector, not the control
    ctl = run(None)                                                 # the mask does not outlive the call that carried it
    assert rel_l2(ctl, rc) < 3e-2 and rel_l2(ctl, rc) < 0.5 * rel_l2(ctl, rm)
    assert rel_l2(run(cmask), rm) < 0.003
    # float (additive) masks and wrong shapes are refused, not reinterpreted
    with pytest.raises(NotImplementedError):
        run(cmask.float())
    with pytest.raises(ValueError):
        run(cmask[:, :8])
    # ragged shape, more keys than one KV tile, V2.3-style per-token timesteps off: against the oracle
    cfg2, w2, m2 = make_dit(dev, heads=2, layers=2, cap=128)
    lat2, ctx2, pos2 = inputs(2, 5, 7, 100, 128)
    mk2 = torch.rand(1, 100, generator=gen) > 0.5
    ctx2 = ctx2.clone()
    ctx2[0, ~mk2[0]] *= 40.0
    sigma = torch.tensor([0.909375])
    ref = dit.x0_model(lat2, ctx2, sigma, pos2, w2, cfg2, context_mask=mk2)
    ref_nomask = dit.x0_model(lat2, ctx2, sigma, pos2, w2, cfg2)
    got = X0Model(m2)(Modality(latent=lat2.to(dev), context=ctx2.to(dev), context_mask=mk2.to(dev), timesteps=sigma.to(dev),
                               positions=pos2.to(dev))).cpu()
    assert rel_l2(got, ref) < 2e-2 and pearson(got, ref) > 0.999 and rel_l2(got, ref) < 0.5 * rel_l2(got, ref_nomask)


@pytest.mark.parametrize("v23", [False, True])
def test_av_masked_text_cross_attention_against_reference_vectors(dev, v23):
    """Modality.context_mask on the AudioVideo engine (19B-style blocks and V2.3): each modality's mask reaches ITS text cross-attention
    (transformer.py:523, 551; audio: head_dim 64, the KM form of the 64-wide kernel) -- against the vectors recorded from the reference's own
    X0Model with both masks (tests/golden/dit_av_tiny.npz), whose no-mask control differs visibly, and cleared again by a mask-less call."""
    import numpy as np
    import os
    from test_oracle_golden import av_masked_case
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, w, video, audio, vm, am = av_masked_case(v23)
    _, _, wq, m = make_av(dev, v23, seed=17 + v23)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_av_tiny.npz"))
    tag = "v23" if v23 else "v1"

    def run(vmask, amask):
        def mod(d, mk):
            return Modality(latent=d["latent"].to(dev), context=d["context"].to(dev), context_mask=None if mk is None else mk.to(dev),
                            timesteps=d["timesteps"].to(dev), positions=d["positions"].to(dev), sigma=d["sigma"].to(dev))
        vx0, ax0 = X0Model(m)(mod(video, vmask), mod(audio, amask))
        return vx0.cpu(), ax0.cpu()
    gv, ga = [torch.from_numpy(z[f"{tag}_masked_{k}_x0"]) for k in ("video", "audio")]
    cv, ca = [torch.from_numpy(z[f"{tag}_masked_control_{k}_x0"]) for k in ("video", "audio")]
    vx0, ax0 = run(vm, am.bool())
    assert rel_l2(vx0, gv) < 5e-3 and rel_l2(ax0, ga) < 5e-3
    assert rel_l2(vx0, gv) < 0.5 * rel_l2(vx0, cv)                  # the masked vector, not the control
    if v23:
        assert rel_l2(ax0, ga) < 0.5 * rel_l2(ax0, ca)              # (v1's audio vectors differ by 5e-3 only: inside the 16-bit tolerance)
    vx0, ax0 = run(None, None)                                      # no mask outlives its call
    assert rel_l2(vx0, cv) < 3e-2 and rel_l2(vx0, cv) < 0.5 * rel_l2(vx0, gv)
    vx0, ax0 = run(vm, None)                                        # one modality masked, the other not
    assert rel_l2(vx0, gv) < 0.5 * rel_l2(vx0, cv)
