"""Parity at the sizes bench.py times ("depth and size never parity-checked"): the fp32 oracle is far too
slow on host cores at these sizes, so the SAME oracle code (torch ops only) is executed in fp32 on the GPU as the checker
-- it stays test infrastructure, never the thing measured.

  * the 48-layer full-width step bench.py's headline times (D = 4096, N = 3456, S = 1024), distinct weights per layer
  * BASELINE config 5 at its real size: a stage-2 step on N = 13 824 tokens, `decode_tiled` of a full-width decoder with
    overlapping spatial AND temporal tiles, and the two-stage pipeline's upscale / re-noise / stage-2 logic against an
    oracle loop (not against itself)
Tolerances as tests/test_parity.py (bf16 operands, fp32 accumulate, fp32 residual stream vs an fp32 oracle)."""
import pytest
import torch

from conftest import rel_l2
from test_parity import inputs, make_dit, make_vae, pearson

pytestmark = pytest.mark.gpu

# accuracy contract of the opt-in fp8-compute mode against the fp32 oracle (48-layer full-width x0)
FP8_COMPUTE_REL_L2, FP8_COMPUTE_PEARSON = 0.10, 0.995       # measured: 0.045-0.073 / 0.9974-0.9990


def dit_weights_on_gpu(cfg, dev, seed):
    """oracle.dit.make_dit_weights' recipe with the tensors drawn on the GPU (19 G parameters are minutes of host RNG);
    2-D linear weights are rounded to bf16 values so oracle and engine see identical numbers."""
    from oracle import dit
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shape in dit.dit_weight_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=dev)
        if name.endswith("_norm.weight"):
            t = 1.0 + 0.1 * t
        elif name.endswith(".bias"):
            t = 0.02 * t
        elif name.endswith("scale_shift_table"):
            t = 0.1 * t
        else:
            t = 0.02 * t
        if name.endswith(".weight") and t.dim() == 2:
            t = t.to(torch.bfloat16).float()
        out[name] = t
    return out


def test_dit_48_layer_step(dev):
    """ONE full denoise-step forward of the headline model: 48 layers, D = 4096, 32 heads, caption 3840, N = 3456, S = 1024."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model
    cfg = dit.DiTConfig(num_layers=48)
    w = dit_weights_on_gpu(cfg, dev, seed=48)
    m = LTXModel(num_layers=48, device=dev)
    m.load_state_dict(w)
    lat, ctx, pos = inputs(9, 16, 24, 1024, 3840, seed=49)
    refs = {}
    for sigma in (1.0, 0.421875):
        ts = torch.tensor([sigma])
        with torch.device(dev), torch.no_grad():
            ref = dit.x0_model(lat.to(dev), ctx.to(dev), ts.to(dev), pos.to(dev), w, cfg).cpu()
        x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
        assert x0.shape == (1, 3456, 128)
        assert rel_l2(x0.cpu(), ref) < 0.012 and pearson(x0.cpu(), ref) > 0.999, sigma
        refs[sigma] = ref
    # BASELINE config 3 as an opt-in fp8-COMPUTE step (fp8 MFMA, e4m3fn weights per output channel + per-token e4m3fn activations in the
    # six projections of every block): same weights, same inputs, against the same fp32 oracle.  Tolerance of THIS mode (stated in
    # DESIGN.md): it must stay far above the reference's own acceptance bar against upstream (Pearson >= 0.95, reference
    # tests/test_parity.py:38).
    del m
    torch.cuda.empty_cache()
    m8 = LTXModel(num_layers=48, device=dev, fp8_compute=True)
    m8.load_state_dict(w)
    assert m8.weight_tensors()["transformer_blocks.0.ff.net.2.weight"].dtype == torch.uint8
    for sigma, ref in refs.items():
        ts = torch.tensor([sigma])
        x8 = X0Model(m8)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
        e, r = rel_l2(x8.cpu(), ref), pearson(x8.cpu(), ref)
        print(f"fp8 compute, 48 layers, sigma {sigma}: rel-L2 {e:.4f}, Pearson {r:.5f}")
        assert e < FP8_COMPUTE_REL_L2 and r > FP8_COMPUTE_PEARSON, (sigma, e, r)
    del w, m8
    torch.cuda.empty_cache()


def test_dit_stage2_size_step(dev):
    """BASELINE config 5, stage 2: 1536x1024x65 -> latent 9 x 32 x 48 = 13 824 tokens; 2 full-width layers, the three stage-2
    steps through the hipGraph against the oracle's loop."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.components import STAGE_2_DISTILLED_SIGMA_VALUES
    cfg, w, m = make_dit(dev, heads=32, layers=2, cap=3840, seed=51)
    f, h, wd = 9, 32, 48
    lat, ctx, pos = inputs(f, h, wd, 1024, 3840, seed=52)
    sig = list(STAGE_2_DISTILLED_SIGMA_VALUES)
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
    assert z.shape == (13824, 128)
    assert rel_l2(z.cpu(), ref[0]) < 0.008 and pearson(z.cpu(), ref[0]) > 0.999


def test_decode_tiled_full_width(dev):
    """`decode_tiled` with the reference's default tiling (512 px / 64 overlap, 64 frames / 24 overlap) and the full-width
    decoder (base_channels 128, default blocks) on a 9 x 16 x 40 latent = 65 x 512 x 1280 px: 3 spatial x 2 temporal
    overlapping tiles and their trapezoid blend, against the oracle's decode_tiled (tiling.py:252-412)."""
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder, TilingConfig, decode_tiled, generate_tile_specs
    cfg = vae.VAEConfig()
    w = vae.make_vae_weights(cfg, seed=61)
    d = SimpleVideoDecoder(device=dev)
    d.load_state_dict(w)
    g = torch.Generator().manual_seed(62)
    z = torch.randn(1, 128, 9, 16, 40, generator=g)
    tc = TilingConfig.default()
    specs = list(generate_tile_specs(z.shape, tc))
    assert len(specs) >= 4 and len({(s.in_w_start, s.in_w_end) for s in specs}) >= 2 and len({(s.in_t_start, s.in_t_end) for s in specs}) >= 2
    zeros = {}          # decode noise fixed to zero on both sides (only the (1 - 0.025) scaling applies, simple_decoder.py:496-498)

    def dec(t, timestep=0.05):
        return d(t, timestep=timestep, noise=zeros.setdefault(tuple(t.shape), torch.zeros(t.shape, device=dev)))

    out = next(decode_tiled(z.to(dev), dec, tc))
    wq = {k: (v.to(torch.bfloat16).float() if (v.dim() == 5 or (v.dim() == 2 and "linear" in k)) else v).to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ref = vae.decode_tiled(z.to(dev), lambda t: vae.decoder_forward(t, wq, cfg, timestep=0.05, noise=None))
    assert out.shape == ref.shape == (1, 3, 65, 512, 1280)
    assert rel_l2(out.cpu(), ref.cpu()) < 0.03 and pearson(out.cpu(), ref.cpu()) > 0.999


def test_two_stage_pipeline_against_oracle_loop(dev):
    """Stage 1 (8 steps at half resolution) -> un_normalize / upscale x2 / normalize -> re-noise to sigma 0.909375 with a
    SUPPLIED noise tensor -> stage 2 (3 steps) against the same sequence written with the oracle's pieces
    (pipelines/distilled.py:394-470); eager and hipGraph paths."""
    from oracle import dit, loop, upscaler as oup
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.pipelines import DistilledConfig, DistilledPipeline
    cfg, w, m = make_dit(dev, heads=2, layers=2, cap=128, seed=71)
    vcfg, vw, d = make_vae(dev, layers=1, seed=72)
    uw = oup.make_upscaler_weights(128, 64, 1, seed=73)
    up = SpatialUpscaler(in_channels=128, mid_channels=64, num_blocks_per_stage=1, device=dev)
    up.load_state_dict(uw)
    uwq = {k: (v.to(torch.bfloat16).float() if v.dim() >= 4 else v) for k, v in uw.items()}
    g = torch.Generator().manual_seed(74)
    f, h1, w1 = 3, 4, 6
    ctx = 0.1 * torch.randn(1, 64, 128, generator=g)
    noise1 = torch.randn(1, f * h1 * w1, 128, generator=g)
    noise2 = torch.randn(1, f * 2 * h1 * 2 * w1, 128, generator=g)
    # ---- oracle sequence
    pos1, pos2 = loop.video_positions(1, f, h1, w1, 24.0), loop.video_positions(1, f, 2 * h1, 2 * w1, 24.0)
    ones1, ones2 = torch.ones(1, f * h1 * w1, 1), torch.ones(1, f * 4 * h1 * w1, 1)
    tok = loop.gaussian_noiser(torch.zeros_like(noise1), ones1, noise1, 1.0)
    tok = loop.denoise_loop_pipeline(tok, ones1, torch.zeros_like(noise1), lambda x, ts, s: dit.x0_model(x, ctx, ts, pos1, w, cfg),
                                     loop.DISTILLED_SIGMA_VALUES)
    lat1 = loop.unpatchify(tok, f, h1, w1)
    mean, std = vw["vae.per_channel_statistics.mean-of-means"].float(), vw["vae.per_channel_statistics.std-of-means"].float()
    lat_up = oup.upscale_latent(lat1, uwq, mean, std, num_blocks=1)
    s0 = float(loop.STAGE_2_DISTILLED_SIGMA_VALUES[0])
    tok2 = loop.gaussian_noiser(loop.patchify(lat_up), ones2, noise2, s0)
    tok2 = loop.denoise_loop_pipeline(tok2, ones2, torch.zeros_like(noise2), lambda x, ts, s: dit.x0_model(x, ctx, ts, pos2, w, cfg),
                                      loop.STAGE_2_DISTILLED_SIGMA_VALUES)
    ref = loop.unpatchify(tok2, f, 2 * h1, 2 * w1)
    # ---- product pipeline
    pipe = DistilledPipeline(m, d, None, spatial_upscaler=up)         # decoder object as the statistics provider, latent out
    for graph in (False, True):
        conf = DistilledConfig(height=256, width=384, num_frames=17, seed=1, use_hip_graph=graph)
        lat = pipe(ctx.to(dev), None, conf, initial_noise=noise1.to(dev), stage2_noise=noise2.to(dev))
        assert lat.shape == ref.shape == (1, 128, 3, 8, 12)
        assert rel_l2(lat.cpu(), ref) < 0.005 and pearson(lat.cpu(), ref) > 0.999, graph


def test_gemm_w8a16_is_bit_identical_to_dequantise_at_load(dev):
    """fp8-RESIDENT weights (BASELINE config 3): the GEMM that expands e4m3fn codes x per-row scale on the way to the
    MFMA gives BIT-IDENTICAL outputs to the bf16 GEMM on weights dequantised at load (reference fp8_loader.py:14-51 arithmetic), on
    the DiT shapes (N = 3456 tokens; to_out / QKV / FFN-up / FFN-down) and a ragged small-M case, for every epilogue the DiT uses."""
    import ltx_2_mlx_amd.kernels as K
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator(device=dev).manual_seed(5)
    for (M, N, Kk) in ((3456, 4096, 4096), (3456, 12288, 4096), (3456, 16384, 4096), (3456, 4096, 16384), (70, 512, 256)):
        a = torch.randn(M, Kk, generator=g, device=dev).to(torch.bfloat16)
        w = torch.randn(N, Kk, generator=g, device=dev) / Kk ** 0.5
        parts = 3 if N % 3 == 0 and N > 256 * 3 else 1              # fused q/k/v: one per-tensor scale per part
        scale = torch.cat([torch.full((N // parts,), float(w[i * (N // parts):(i + 1) * (N // parts)].abs().max() / 448.0), device=dev) for i in range(parts)])
        codes = (w / scale[:, None]).to(torch.float8_e4m3fn)
        wdq = torch.cat([K.dequant_fp8(codes[i * (N // parts):(i + 1) * (N // parts)].view(torch.uint8), float(scale[i * (N // parts)])) for i in range(parts)])
        assert torch.equal(wdq.float(), (codes.float() * scale[:, None]).to(torch.bfloat16).float())         # the load-time kernel is f32(code)*scale -> bf16
        bias = torch.randn(N, generator=g, device=dev)
        for epi in (nv.EPI_BF16, nv.EPI_GELU_BF16, nv.EPI_F32):
            ref = K.gemm(a, wdq, bias, epilogue=epi)
            out = K.gemm_w8a16(a, codes.view(torch.uint8), scale, bias, epilogue=epi)
            assert torch.equal(out, ref), (M, N, Kk, epi)
        gate = torch.randn(N, generator=g, device=dev)
        x0 = torch.randn(M, N, generator=g, device=dev)
        xr, xo = x0.clone(), x0.clone()
        K.gemm(a, wdq, bias, epilogue=nv.EPI_RESID_GATE_F32, out=xr, gate_table=gate)
        K.gemm_w8a16(a, codes.view(torch.uint8), scale, bias, epilogue=nv.EPI_RESID_GATE_F32, out=xo, gate_table=gate)
        assert torch.equal(xo, xr), (M, N, Kk, "resid")


def test_fp8_resident_model_matches_dequantised_model(dev, tmp_path):
    """load_transformer_weights(use_fp8=True, fp8_resident=True): the attention / feed-forward projections stay float8_e4m3fn + scale in
    HBM; the 2-layer full-width model's x0 is BIT-IDENTICAL to the same checkpoint dequantised at load, and its weights take ~half the bytes."""
    from safetensors.torch import save_file
    from oracle import dit
    from ltx_2_mlx_amd.loader import load_transformer_weights
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model
    cfg = dit.DiTConfig(num_layers=2)
    w = dit_weights_on_gpu(cfg, dev, seed=81)
    ck = {}
    for k, v in w.items():
        full = "model.diffusion_model." + k
        if k.endswith(".weight") and v.dim() == 2 and "transformer_blocks" in k:
            scale = float(v.abs().max() / 448.0)
            ck[full] = (v / scale).to(torch.float8_e4m3fn).cpu()
            ck[full + "_scale"] = torch.tensor(scale)
        else:
            ck[full] = (v.to(torch.bfloat16) if (k.endswith(".weight") and v.dim() == 2) else v).cpu()
    path = str(tmp_path / "l2_fp8.safetensors")
    save_file(ck, path)
    del w, ck
    lat, ctx, pos = inputs(9, 16, 24, 1024, 3840, seed=82)
    outs, nbytes = [], []
    for resident in (False, True):
        m = LTXModel(num_layers=2, device=dev)
        load_transformer_weights(m, path, strict=True, use_fp8=True, fp8_resident=resident)
        m.set_option("fold_norms", 0)       # the same PROGRAM on both: the folded pre-norm (round 6) has kernels for dense 16-bit weights only, the fp8-resident model keeps the norm pass
        nbytes.append(sum(t.numel() * t.element_size() for t in m.weight_tensors().values()))
        x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=torch.tensor([0.725], device=dev), positions=pos.to(dev)))
        outs.append(x0.clone())
        del m
        torch.cuda.empty_cache()
    assert torch.equal(outs[0], outs[1])
    assert nbytes[1] < 0.62 * nbytes[0]


def av_weights_on_gpu(cfg, dev, seed):
    """oracle.dit_av.make_av_weights' recipe drawn on the GPU (the 48-layer AudioVideo model is ~22 G parameters); 2-D linear
    weights are rounded to bf16 values so oracle and engine see identical numbers."""
    from oracle import dit_av
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shape in dit_av.av_weight_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=dev)
        if name.endswith("_norm.weight"):
            t = 1.0 + 0.1 * t
        elif name.endswith(".bias"):
            t = 0.02 * t
        elif "scale_shift_table" in name:
            t = 0.1 * t
        else:
            t = 0.02 * t
        if name.endswith(".weight") and t.dim() == 2:
            t = t.to(torch.bfloat16).float()
        out[name] = t
    return out


def test_fp8_compute_trajectory_and_outlier_channels(dev):
    """fp8 compute (BASELINE config 3) beyond one forward (VERDICT r3 weak #1a): (i) the whole 8-step distilled trajectory of the 48-layer
    full-width model through the hipGraph against the fp32 oracle's loop -- the per-step error feeds the next step through the Euler update;
    (ii) the same with OUTLIER input channels (x20 on 0.1 % of the input channels of every attention / feed-forward projection, the case a
    per-token activation scale + per-output-channel weight scale handles worst: the outlier sets the scale of its whole row).  The bf16 engine
    runs the same two trajectories as the yardstick.  Bars: the fp8 mode's single-forward bars hold for the END of the trajectory too (rel-L2 < 0.10,
    Pearson > 0.995; measured on MI355X: 0.057 / 0.9984 and, with the outlier channels, 0.062 / 0.9981; bf16: 0.0022 / 1.0000 both times --
    the reference's own acceptance bar against upstream is Pearson >= 0.95 for ONE forward, reference tests/test_parity.py:38)."""
    import re
    from oracle import dit, loop
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
    from ltx_2_mlx_amd.model.transformer import LTXModel
    cfg = dit.DiTConfig(num_layers=48)
    f, h, wd = 9, 16, 24
    lat, ctx, pos = inputs(f, h, wd, 1024, 3840, seed=149)
    sig = DISTILLED_SIGMA_VALUES
    lin = re.compile(r"^transformer_blocks\.\d+\.(attn1|attn2)\.(to_q|to_k|to_v|to_out\.0)\.weight$|^transformer_blocks\.\d+\.ff\.net\.(0\.proj|2)\.weight$")
    for outliers in (False, True):
        w = dit_weights_on_gpu(cfg, dev, seed=148)
        if outliers:
            g = torch.Generator(device=dev).manual_seed(150)
            for name, t in w.items():
                if lin.match(name):
                    cols = torch.randperm(t.shape[1], generator=g, device=dev)[:max(1, t.shape[1] // 1000)]
                    t[:, cols] *= 20.0
                    w[name] = t.to(torch.bfloat16).float()
        with torch.device(dev), torch.no_grad():
            ctx_g, pos_g = ctx.to(dev), pos.to(dev)
            ref = loop.denoise_loop_cli(loop.unpatchify(lat.to(dev), f, h, wd), lambda tok, s: dit.x0_model(tok, ctx_g, torch.tensor([s]), pos_g, w, cfg), sig)
            ref = loop.patchify(ref)[0].cpu()
        assert torch.isfinite(ref).all()
        res = {}
        for fp8 in (False, True):
            m = LTXModel(num_layers=48, device=dev, fp8_compute=fp8)
            m.load_state_dict(w)
            m.prepare(ctx.to(dev), pos.to(dev))
            z = lat[0].to(dev).contiguous()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.capture_denoise_graph(z, sig)
                m.replay_denoise_graph()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            res[fp8] = (rel_l2(z.cpu(), ref), pearson(z.cpu(), ref))
            del m
            torch.cuda.empty_cache()
        print(f"8-step trajectory, 48 layers, outlier channels {outliers}: bf16 rel-L2 {res[False][0]:.4f} Pearson {res[False][1]:.5f} | "
              f"fp8 compute rel-L2 {res[True][0]:.4f} Pearson {res[True][1]:.5f}")
        assert res[False][0] < 1e-2 and res[False][1] > 0.999, res            # measured 2.2e-3
        assert res[True][0] < FP8_COMPUTE_REL_L2 and res[True][1] > FP8_COMPUTE_PEARSON, res
        del w
        torch.cuda.empty_cache()


@pytest.mark.parametrize("v23", [False, True])
def test_av_fp8_compute_step(dev, v23):
    """AudioVideo + fp8 compute: the video stream's projections on the fp8 MFMA (the audio stream and the cross-modal attention stay 16-bit), two
    full-width layers, joint x0 against the fp32 oracle executed on the GPU; the fp8 mode's accuracy bars."""
    from oracle import dit_av, loop
    from test_parity import to_modality
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, X0Model
    cfg = dit_av.AVConfig(num_attention_heads=32, attention_head_dim=128, audio_heads=32, audio_head_dim=64, num_layers=2,
                          caption_channels=None if v23 else 3840, cross_attention_adaln=v23, apply_gated_attention=v23)
    w = dit_av.make_av_weights(cfg, seed=171 + v23)
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=32, attention_head_dim=128, num_layers=2, caption_channels=cfg.caption_channels,
                 cross_attention_adaln=v23, apply_gated_attention=v23, audio_attention_heads=32, device=dev, fp8_compute=True)
    m.load_state_dict(w)
    assert m.weight_tensors()["transformer_blocks.0.attn1.to_qkv.weight"].dtype == torch.uint8                 # video stream: e4m3fn codes
    assert m.weight_tensors()["transformer_blocks.0.audio_attn1.to_qkv.weight"].dtype == torch.bfloat16        # audio stream: 16-bit
    g = torch.Generator().manual_seed(172)
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
    ev, pv, ea, pa = rel_l2(vx0.cpu(), rv.cpu()), pearson(vx0.cpu(), rv.cpu()), rel_l2(ax0.cpu(), ra.cpu()), pearson(ax0.cpu(), ra.cpu())
    print(f"AudioVideo fp8 compute (v23={v23}): video rel-L2 {ev:.4f} Pearson {pv:.5f} | audio rel-L2 {ea:.4f} Pearson {pa:.5f}")
    assert ev < FP8_COMPUTE_REL_L2 and pv > FP8_COMPUTE_PEARSON and ea < FP8_COMPUTE_REL_L2 and pa > FP8_COMPUTE_PEARSON


def test_av_48_layer_step_v23(dev):
    """The step bench.py times as `ltx23_audiovideo_ms_per_step` (BASELINE config 4): 48-layer full-width LTX-2.3 AudioVideo model
    (video 32 x 128 + audio 32 x 64 heads, 9-row AdaLN, prompt AdaLN, per-head gates, cross-modal attention), 3456 video + 68
    audio tokens, S = 1024, distinct weights per layer: one joint x0 evaluation against the fp32 oracle executed on the GPU."""
    from oracle import dit_av, loop
    from test_parity import to_modality
    from ltx_2_mlx_amd.model.transformer import LTXModel, LTXModelType, X0Model
    cfg = dit_av.AVConfig(num_attention_heads=32, attention_head_dim=128, audio_heads=32, audio_head_dim=64, num_layers=48,
                          caption_channels=None, cross_attention_adaln=True, apply_gated_attention=True)
    w = av_weights_on_gpu(cfg, dev, seed=148)
    m = LTXModel(model_type=LTXModelType.AudioVideo, num_attention_heads=32, attention_head_dim=128, num_layers=48, caption_channels=None,
                 cross_attention_adaln=True, apply_gated_attention=True, audio_attention_heads=32, device=dev)
    m.load_state_dict(w)
    g = torch.Generator().manual_seed(149)
    f, h, wd, Ta, S = 9, 16, 24, 68, 1024
    s = torch.tensor([0.725])
    video = dict(latent=torch.randn(1, f * h * wd, 128, generator=g), context=0.1 * torch.randn(1, S, cfg.inner_dim, generator=g),
                 timesteps=s, sigma=s, positions=loop.video_positions(1, f, h, wd, 25.0))
    audio = dict(latent=torch.randn(1, Ta, 128, generator=g), context=0.1 * torch.randn(1, S, cfg.audio_inner_dim, generator=g),
                 timesteps=s, sigma=s, positions=dit_av.audio_positions(1, Ta))
    vx0, ax0 = X0Model(m)(to_modality(video, dev), to_modality(audio, dev))
    with torch.device(dev), torch.no_grad():
        rv, ra = dit_av.av_x0_model({k: t.to(dev) for k, t in video.items()}, {k: t.to(dev) for k, t in audio.items()}, w, cfg)
    assert vx0.shape == (1, 3456, 128) and ax0.shape == (1, 68, 128)
    assert rel_l2(vx0.cpu(), rv.cpu()) < 0.012 and pearson(vx0.cpu(), rv.cpu()) > 0.999
    assert rel_l2(ax0.cpu(), ra.cpu()) < 0.008 and pearson(ax0.cpu(), ra.cpu()) > 0.999
    del w, m
    torch.cuda.empty_cache()



def test_fold_norms_agrees_and_graph_is_bit_identical(dev):
    """Round 6: the text cross-attention's plain RMS pre-norm folded around attn1.to_out / attn2.to_q (engine option fold_norms, default 1).  At the headline
    geometry (D = 4096, N = 3456) with 3 layers: the 8-step loop with the fold and without it (0: round 5's norm pass) against the fp32 oracle's loop; the two
    agree with each other far inside the oracle gate, the fold really runs another program (not bit-equal), and the captured loop equals the eager steps bit for bit."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, m = make_dit(dev, heads=32, layers=3, cap=3840, seed=61)
    f, h, wd = 9, 16, 24
    lat, ctx, pos = inputs(f, h, wd, 256, 3840, seed=62)
    sig = DISTILLED_SIGMA_VALUES
    wg = {k: v.to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ctx_g, pos_g = ctx.to(dev), pos.to(dev)
        ref = loop.denoise_loop_cli(loop.unpatchify(lat.to(dev), f, h, wd),
                                    lambda tok, s: dit.x0_model(tok, ctx_g, torch.tensor([s]), pos_g, wg, cfg), sig)
        ref = loop.patchify(ref).cpu()[0]
    C, P = ctx.to(dev), pos.to(dev)

    def eager(level):
        m.set_option("fold_norms", level)
        y = lat[0].to(dev).contiguous()
        for i in range(len(sig) - 1):
            mod = Modality(latent=y[None], context=C, context_mask=None, timesteps=torch.tensor([sig[i]], device=dev), positions=P)
            m.denoise_step_(y, mod, sig[i], sig[i + 1])
        return y

    outs = {lv: eager(lv) for lv in (0, 1)}
    for lv, y in outs.items():
        assert rel_l2(y.cpu(), ref) < 0.008 and pearson(y.cpu(), ref) > 0.999, lv
    assert rel_l2(outs[1].cpu(), outs[0].cpu()) < 5e-3
    assert not torch.equal(outs[1], outs[0])
    assert torch.equal(eager(1), outs[1])
    for _ in range(2):          # captured loop == eager steps, bit for bit (twice: replays are re-entrant)
        z = lat[0].to(dev).contiguous()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.capture_denoise_graph(z, sig)
            m.replay_denoise_graph()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(z, outs[1])
    with pytest.raises(ValueError):
        m.set_option("fold_norms", 2)


def test_text_qnorm_fold_falls_back_at_20_heads(dev):
    """D = 2560 (20 heads x 128) gives q_ss_ld = D / 64 = 40, not a multiple of 16 -- the attention kernel's row-scale form rejects it, so
    ltx2_dit_prepare must keep the q_norm pass for such models (text_qfold_ok gates on D % 1024).  N >= 1024 so the query projection runs on the
    4-wave kernel (the route that would otherwise enable the fold).  Against the fp32 oracle."""
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import Modality
    cfg, w, m = make_dit(dev, heads=20, layers=1, cap=128, seed=71)
    lat, ctx, pos = inputs(3, 16, 24, 64, 128, seed=72)
    sigma = torch.tensor([0.725])
    wg = {k: v.to(dev) for k, v in w.items()}
    with torch.device(dev), torch.no_grad():
        ref = dit.velocity_model(lat.to(dev), ctx.to(dev), sigma.to(dev), pos.to(dev), wg, cfg).cpu()
    v = m(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=sigma.to(dev), positions=pos.to(dev)))
    assert v.shape == (1, 1152, 128)
    assert rel_l2(v.cpu(), ref) < 0.012 and pearson(v.cpu(), ref) > 0.999
