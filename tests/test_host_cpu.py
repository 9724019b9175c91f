"""CPU-only tests (run with -m "not gpu"): the oracle against the reference's own closed-form
known answers, the host-side logic of the product against the oracle, the C-ABI symbol table,
and the world_size-2 gloo path of the prompt sharding / weight broadcast."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- oracle vs reference known answers
def test_sigma_tables_match_reference_constants():
    # reference components/schedulers.py:236-253, asserted by its tests/test_scheduler.py:103-144
    from oracle import loop
    from ltx_2_mlx_amd.components import DISTILLED_SIGMA_VALUES, STAGE_2_DISTILLED_SIGMA_VALUES
    assert loop.DISTILLED_SIGMA_VALUES == [1.0, 0.99375, 0.9875, 0.98125, 0.975, 0.909375, 0.725, 0.421875, 0.0]
    assert loop.STAGE_2_DISTILLED_SIGMA_VALUES == [0.909375, 0.725, 0.421875, 0.0]
    assert DISTILLED_SIGMA_VALUES == loop.DISTILLED_SIGMA_VALUES
    assert STAGE_2_DISTILLED_SIGMA_VALUES == loop.STAGE_2_DISTILLED_SIGMA_VALUES
    assert len(DISTILLED_SIGMA_VALUES) == 9 and len(STAGE_2_DISTILLED_SIGMA_VALUES) == 4
    assert DISTILLED_SIGMA_VALUES[-1] == 0.0 and all(a >= b for a, b in zip(DISTILLED_SIGMA_VALUES, DISTILLED_SIGMA_VALUES[1:]))


def test_post_process_and_timesteps_known_answers():
    # reference tests/test_pipelines.py:193-266
    from oracle import loop
    from ltx_2_mlx_amd.pipelines import post_process_latent, timesteps_from_mask
    den = torch.tensor([[[1.0, 2.0], [3.0, 4.0]]])
    clean = torch.tensor([[[5.0, 6.0], [7.0, 8.0]]])
    for fn in (loop.post_process_latent, post_process_latent):
        assert torch.allclose(fn(den, torch.ones_like(den), clean), den)
        assert torch.allclose(fn(den, torch.zeros_like(den), clean), clean)
        assert torch.allclose(fn(torch.tensor([[[1.0, 2.0]]]), torch.tensor([[[1.0, 0.0]]]), torch.tensor([[[5.0, 6.0]]])),
                              torch.tensor([[[1.0, 6.0]]]))
    for fn in (loop.timesteps_from_mask, timesteps_from_mask):
        assert torch.allclose(fn(torch.ones(1, 10), 0.5), torch.full((1, 10), 0.5))
        assert torch.allclose(fn(torch.tensor([[1.0, 0.5, 0.0]]), 2.0), torch.tensor([[2.0, 1.0, 0.0]]))


def test_latent_shape_known_answer():
    # reference tests/test_pipelines.py:269-297: 97 x 480 x 704 -> 13 x 15 x 22
    from oracle import loop
    from ltx_2_mlx_amd.types import VideoLatentShape, VideoPixelShape
    assert loop.latent_shape_from_pixels(97, 480, 704) == (13, 15, 22)
    s = VideoLatentShape.from_pixel_shape(VideoPixelShape(1, 97, 480, 704, 24.0))
    assert (s.batch, s.channels, s.frames, s.height, s.width) == (1, 128, 13, 15, 22)
    assert loop.latent_shape_from_pixels(65, 512, 768) == (9, 16, 24)        # BASELINE main config: N = 3456


def test_patchify_roundtrip_and_positions():
    from oracle import loop
    from ltx_2_mlx_amd.components import VideoLatentPatchifier
    from ltx_2_mlx_amd.conditioning import VideoLatentTools
    from ltx_2_mlx_amd.types import VideoLatentShape
    x = torch.randn(1, 128, 3, 4, 5)
    p = VideoLatentPatchifier(1)
    shp = VideoLatentShape.from_shape(x.shape)
    tok = p.patchify(x)
    assert tok.shape == (1, 60, 128) and torch.equal(tok, loop.patchify(x))
    assert torch.equal(p.unpatchify(tok, shp), x) and torch.equal(loop.unpatchify(tok, 3, 4, 5), x)
    st = VideoLatentTools(p, shp, fps=24.0).create_initial_state()
    assert st.latent.shape == (1, 60, 128) and st.denoise_mask.shape == (1, 60, 1) and st.positions.shape == (1, 3, 60, 2)
    assert torch.allclose(st.positions, loop.video_positions(1, 3, 4, 5, 24.0))
    # causal fix: first latent frame covers [0, 1/24) s; second [1/24, 9/24)
    assert torch.allclose(st.positions[0, 0, 0], torch.tensor([0.0, 1 / 24]))
    assert torch.allclose(st.positions[0, 0, 20], torch.tensor([1 / 24, 9 / 24]))
    with pytest.raises(ValueError):
        VideoLatentTools(p, shp, fps=24.0).create_initial_state(initial_latent=torch.zeros(1, 128, 3, 4, 6))


def test_ltx2_scheduler_properties():
    from oracle import loop
    from ltx_2_mlx_amd.components import LTX2Scheduler
    for steps in (2, 8, 30):
        s = LTX2Scheduler().execute(steps)
        assert torch.allclose(s, loop.ltx2_scheduler(steps))
        assert s.shape == (steps + 1,) and abs(float(s[0]) - 1.0) < 1e-6 and s[-1] == 0.0
        assert bool((s[:-1] >= s[1:]).all())
        assert abs(float(s[-2]) - 0.1) < 1e-6                         # stretched to terminal 0.1
    lat = torch.zeros(1, 128, 9, 16, 24)
    assert torch.allclose(LTX2Scheduler().execute(8, latent=lat), loop.ltx2_scheduler(8, tokens=3456))


def test_rope_tables_layout():
    """SPLIT RoPE: 2 identity slots at the FRONT, slot = f*3 + d, head h gets slots [64h, 64h+64)."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.model.transformer import rope_tables_token_major
    heads, D = 32, 4096
    pos = loop.video_positions(1, 2, 3, 4, 24.0)
    cos, sin = dit.rope_split_tables(pos, D, heads, 10000.0, [20, 2048, 2048])
    assert cos.shape == (1, 32, 24, 64)
    assert torch.all(cos[0, 0, :, :2] == 1) and torch.all(sin[0, 0, :, :2] == 0)     # 4096/2 - 3*682 = 2 pad
    grid = dit.rope_freq_grid(10000.0, 3, D)
    assert grid.shape == (682,) and abs(float(grid[0]) - torch.pi / 2) < 1e-6
    tok = 13
    mid = (pos[0, :, tok, 0] + pos[0, :, tok, 1]) / 2
    frac = mid / torch.tensor([20.0, 2048.0, 2048.0])
    # slot 2 + (f*3 + d) -> head 0, index 2 + 3f + d
    for f_idx, d_idx in [(0, 0), (0, 2), (5, 1)]:
        expect = torch.cos(grid[f_idx] * (frac[d_idx] * 2 - 1))
        assert abs(float(cos[0, 0, tok, 2 + 3 * f_idx + d_idx]) - float(expect)) < 1e-6
    ct, st = rope_tables_token_major(pos, D, heads, 10000.0, [20, 2048, 2048])
    assert torch.equal(ct, cos[0].permute(1, 0, 2).reshape(24, D // 2))
    assert torch.equal(st, sin[0].permute(1, 0, 2).reshape(24, D // 2))
    with pytest.raises(ValueError):
        rope_tables_token_major(pos, D, heads, 10000.0, [20, 2048])


def test_split_rope_is_a_rotation():
    from oracle import dit, loop
    pos = loop.video_positions(1, 2, 2, 2, 24.0)
    cos, sin = dit.rope_split_tables(pos, 256, 2, 10000.0, [20, 2048, 2048])
    x = torch.randn(1, 8, 256)
    y = dit.apply_split_rope(x, cos, sin)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), rtol=1e-5)                   # norm preserving
    assert torch.allclose(dit.apply_split_rope(y, cos, -sin), x, atol=1e-5)            # inverse rotation


def test_oracle_conv3d_matches_tap_loop():
    """conv3d_simple == the reference's 3-tap conv2d accumulation with explicit padding."""
    import torch.nn.functional as F
    from oracle import vae
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 3, 5, 6, generator=g)
    w = torch.randn(7, 4, 3, 3, 3, generator=g)
    b = torch.randn(7, generator=g)
    for causal in (False, True):
        xp = torch.cat([x[:, :, :, 1:2].flip(3), x, x[:, :, :, -2:-1].flip(3)], 3)
        xp = torch.cat([xp[..., 1:2].flip(4), xp, xp[..., -2:-1].flip(4)], 4)
        xp = torch.cat([xp[:, :, :1]] * (2 if causal else 1) + [xp] + ([] if causal else [xp[:, :, -1:]]), 2)
        out = None
        for kt in range(3):
            xs = xp[:, :, kt:kt + 3].permute(0, 2, 1, 3, 4).reshape(3, 4, 7, 8)
            o = F.conv2d(xs, w[:, :, kt]).reshape(1, 3, 7, 5, 6).permute(0, 2, 1, 3, 4)
            out = o if out is None else out + o
        out = out + b[None, :, None, None, None]
        assert torch.allclose(vae.conv3d_simple(x, w, b, causal), out, atol=1e-4)


def test_oracle_unpatchify_and_d2s_orders():
    from oracle import vae
    # unpatchify: packing (c, p, r_w, r_h): channel c*16 + rw*4 + rh -> pixel (h*4+rh, w*4+rw)
    x = torch.zeros(1, 48, 1, 2, 2)
    x[0, 1 * 16 + 2 * 4 + 3, 0, 1, 0] = 5.0
    y = vae.unpatchify(x, 4, 1)
    assert y.shape == (1, 3, 1, 8, 8) and y[0, 1, 0, 1 * 4 + 3, 0 * 4 + 2] == 5.0 and y.abs().sum() == 5.0
    # depth to space: channel ((c*ft + a)*fh + b)*fw + d -> (t*ft+a, h*fh+b, w*fw+d)
    z = torch.zeros(1, 16, 2, 2, 2)
    z[0, ((1 * 2 + 1) * 2 + 0) * 2 + 1, 1, 0, 1] = 3.0
    d = vae.depth_to_space(z, 2, (2, 2, 2))
    assert d[0, 1, 1 * 2 + 1, 0 * 2 + 0, 1 * 2 + 1] == 3.0 and d.abs().sum() == 3.0


def test_decode_chunk_walk_and_blend():
    from oracle import vae
    assert vae.temporal_chunks(9) == [(0, 7), (5, 9)]
    assert vae.temporal_chunks(7) == [(0, 7)]
    assert vae.temporal_chunks(13) == [(0, 7), (5, 12), (10, 13)]
    assert vae.latent_t_to_pixel_t(9) == 65 and vae.latent_t_to_pixel_t(2) == 9
    a = torch.zeros(1, 3, 49, 2, 2)
    b = torch.ones(1, 3, 25, 2, 2)
    v = vae.blend_chunks([a, b], 9)
    assert v.shape[2] == 65
    assert torch.all(v[:, :, :40] == 0) and torch.all(v[:, :, 49:] == 1)
    assert torch.allclose(v[0, 0, 40:49, 0, 0], torch.linspace(0, 1, 9))


def test_tile_specs_and_masks_match_oracle():
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import TilingConfig, compute_trapezoidal_mask_1d, generate_tile_specs
    specs = generate_tile_specs((1, 128, 9, 32, 48), TilingConfig.default())
    ref = vae.tile_specs((1, 128, 9, 32, 48))
    assert len(specs) == len(ref) == 24                                   # 2 x 3 x 4 tiles
    for s, r in zip(specs, ref):
        assert (s.in_t_start, s.in_t_end) == r["in_t"] and (s.in_h_start, s.in_h_end) == r["in_h"] and (s.in_w_start, s.in_w_end) == r["in_w"]
        assert (s.out_t_start, s.out_t_end) == r["out_t"] and (s.ramp_h_left, s.ramp_h_right) == r["ramp_h"]
    for args in [(10, 3, 2, False), (10, 3, 0, True), (5, 9, 9, False), (64, 0, 24, True)]:
        assert torch.allclose(compute_trapezoidal_mask_1d(*args), vae.trapezoid_mask_1d(*args))
    m = compute_trapezoidal_mask_1d(6, 2, 0, False)
    assert torch.allclose(m, torch.tensor([1 / 3, 2 / 3, 1, 1, 1, 1]))
    with pytest.raises(ValueError):
        compute_trapezoidal_mask_1d(0, 0, 0)


def test_config_validation_matches_reference_errors():
    from ltx_2_mlx_amd.model.video_vae import SpatialTilingConfig, TemporalTilingConfig
    from ltx_2_mlx_amd.pipelines import DistilledConfig
    with pytest.raises(ValueError, match="8\\*k \\+ 1"):
        DistilledConfig(num_frames=64)
    with pytest.raises(ValueError, match="divisible by 64"):
        DistilledConfig(height=500, width=704)
    assert DistilledConfig(height=512, width=768, num_frames=65)._get_tiling_config() is None      # 3456 voxels
    assert DistilledConfig(height=1024, width=1536, num_frames=65)._get_tiling_config() is not None  # 13824 > 4000
    with pytest.raises(ValueError):
        SpatialTilingConfig(32)
    with pytest.raises(ValueError):
        TemporalTilingConfig(16, 16)


def test_euler_oracle_known_answer():
    from oracle import loop
    x = torch.tensor([[2.0, -1.0]])
    x0 = torch.tensor([[1.0, 1.0]])
    # v = (x - x0)/sigma = [2, -4]; x + v*(0.25-0.5) = [1.5, 0]
    assert torch.allclose(loop.euler_step(x, x0, 0.5, 0.25), torch.tensor([[1.5, 0.0]]))
    assert torch.allclose(loop.euler_step(x, x0, 0.5, 0.0), x0)            # stepping to sigma 0 lands on x0
    with pytest.raises(ValueError, match="Sigma can't be 0.0"):
        loop.euler_step(x, x0, 0.0, 0.0)


def test_oracle_dit_invariances():
    """Size-independent properties: per-token uniform timesteps == scalar timestep; zero gates make a
    block the identity on the residual stream."""
    from oracle import dit, loop
    cfg = dit.DiTConfig(num_attention_heads=2, num_layers=1, caption_channels=64)
    w = dit.make_dit_weights(cfg, 0)
    lat = torch.randn(1, 12, 128)
    ctx = torch.randn(1, 5, 64) * 0.1
    pos = loop.video_positions(1, 1, 3, 4, 24.0)
    a = dit.x0_model(lat, ctx, torch.tensor([0.7]), pos, w, cfg)
    b = dit.x0_model(lat, ctx, torch.full((1, 12, 1), 0.7), pos, w, cfg)
    assert torch.allclose(a, b, atol=1e-5)


# ---------------------------------------------------------------- C ABI
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ltx2hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ltx2_[a-z0-9_]+)\s*\(", txt)))


def test_capi_exports_every_declared_symbol():
    from ltx_2_mlx_amd import _native as nv
    if not os.path.exists(nv.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(nv.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libltx2hip.so does not export {name}"
    assert sorted(nv.exported_symbols()) == declared          # python binding table == header
    assert nv.lib().ltx2_abi_version() == 1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ltx-2-mlx_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "reference" not in dp


def test_no_cpu_fallback():
    from ltx_2_mlx_amd import kernels as K
    from ltx_2_mlx_amd.model.transformer import LTXModel
    with pytest.raises(RuntimeError):
        K.gemm(torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):
        LTXModel(num_attention_heads=2, num_layers=1, device="cpu")
    # every other model class on the path refuses a CPU device the same way
    from ltx_2_mlx_amd.model.text_encoder import Embeddings1DConnector, GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2
    from ltx_2_mlx_amd.model.upscaler import SpatialUpscaler
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder
    from ltx_2_mlx_amd.model.video_vae_encoder import SimpleVideoEncoder
    for ctor in (Embeddings1DConnector, GemmaFeaturesExtractorProjLinear, GemmaFeaturesExtractorV2, SpatialUpscaler, SimpleVideoDecoder,
                 SimpleVideoEncoder):
        with pytest.raises(RuntimeError):
            ctor(device="cpu")


# ---------------------------------------------------------------- multi-process (gloo, world_size 2)
def test_shard_units():
    from ltx_2_mlx_amd.distributed import shard_units
    assert shard_units(8, 0, 8) == [0] and shard_units(8, 7, 8) == [7]
    assert shard_units(8, 1, 2) == [1, 3, 5, 7]
    got = sorted(sum((shard_units(11, r, 4) for r in range(4)), []))
    assert got == list(range(11))
    with pytest.raises(ValueError):
        shard_units(4, 4, 4)


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
from ltx_2_mlx_amd.distributed import init_distributed, broadcast_tensors, shard_units, max_over_ranks, barrier
rank, world, _ = init_distributed("gloo")
g = torch.Generator().manual_seed(100 + rank)          # different content per rank before the broadcast
t = {{"a.weight": torch.randn(300, 7, generator=g).to(torch.bfloat16), "b.bias": torch.randn(11, generator=g),
     "c.weight": torch.randn(5000, generator=g).to(torch.bfloat16), "d.table": torch.randn(6, 16, generator=g)}}
n = broadcast_tensors(t, src=0, bucket_bytes=4096)
g0 = torch.Generator().manual_seed(100)
ref = {{"a.weight": torch.randn(300, 7, generator=g0).to(torch.bfloat16), "b.bias": torch.randn(11, generator=g0),
       "c.weight": torch.randn(5000, generator=g0).to(torch.bfloat16), "d.table": torch.randn(6, 16, generator=g0)}}
assert all(torch.equal(t[k], ref[k]) for k in t), "broadcast mismatch"
assert n >= 3
units = shard_units(5, rank, world)
assert units == ([0, 2, 4] if rank == 0 else [1, 3])
m = max_over_ranks(float(rank + 1), device=torch.device("cpu"))
assert m == 2.0
barrier()
open(os.path.join({out!r}, "ok_%d" % rank), "w").write("OK")
"""


def test_gloo_world2_broadcast_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_save_video_command_and_png_fallback(tmp_path, monkeypatch):
    """CLI file output (reference scripts/generate.py:2153-2226): the ffmpeg filter chain / encoder settings, and
    the PNG-frame fallback used on boxes without an ffmpeg binary."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("ltx2_generate_cli", os.path.join(ROOT, "scripts", "generate.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.video_filters(24, 1.0) == []
    assert gen.video_filters(24, 2.0) == ["setpts=0.5*PTS"]
    assert gen.video_filters(48, 0.5) == ["setpts=2.0*PTS", "minterpolate=fps=48:mi_mode=mci:mc_mode=aobmc:me_mode=bidir:vsbmc=1"]
    cmd = gen.ffmpeg_command(768, 512, "out.mp4", fps=48)
    assert cmd[:2] == ["ffmpeg", "-y"] and "768x512" in cmd and cmd[cmd.index("-framerate") + 1] == "24"
    assert cmd[-9:] == ["-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", "18", "-loglevel", "error", "out.mp4"]
    frames = (np.arange(3 * 8 * 16 * 3) % 251).astype(np.uint8).reshape(3, 8, 16, 3)
    monkeypatch.setattr("shutil.which", lambda name: None)
    out = gen.save_video(frames, str(tmp_path / "clip.mp4"))
    from PIL import Image
    files = sorted(os.listdir(out))
    assert files == ["frame_0000.png", "frame_0001.png", "frame_0002.png"]
    assert np.array_equal(np.asarray(Image.open(os.path.join(out, files[1]))), frames[1])
    with pytest.raises(ValueError):
        gen.save_video(frames.astype(np.float32), str(tmp_path / "bad.mp4"))


def test_generate_video_keeps_the_reference_keyword_surface(tmp_path, capsys):
    """Drop-in boundary: `generate_video` takes the reference's parameters, in order, with the reference's
    defaults (tests/golden/generate_video_signature.json, written from /root/reference/scripts/generate.py:933-997 by
    tools/pin_generate_signature.py); MI355X extras are keyword-only and come after them.  Out-of-path options raise only
    when they are moved off their reference default."""
    import inspect
    import json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import generate
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "generate_video_signature.json")))["params"]
    ours = list(inspect.signature(generate.generate_video).parameters.values())
    assert [p.name for p in ours[:len(ref)]] == [r["name"] for r in ref]
    for p, r in zip(ours, ref):
        assert p.kind == inspect.Parameter.POSITIONAL_OR_KEYWORD
        assert (p.default is inspect.Parameter.empty) == r["required"], p.name
        if not r["required"]:
            assert p.default == r["default"] and type(p.default) is type(r["default"]), (p.name, p.default, r["default"])
    assert all(p.kind == inspect.Parameter.KEYWORD_ONLY for p in ours[len(ref):])
    # every reference default is accepted as given: with use_gemma=True (the default) and no Gemma weights on disk the
    # reference prints an error and returns None (:1085-1092) -- so does this one, before touching the GPU
    kw = {r["name"]: r["default"] for r in ref if not r["required"]}
    kw["output_path"] = str(tmp_path / "gens" / "o.mp4")
    assert generate.generate_video("a prompt", **kw) is None
    assert "Gemma weights not found" in capsys.readouterr().out and os.path.isdir(tmp_path / "gens")
    # an out-of-path option off its default is refused by name; frames / resolution errors keep the reference's wording
    with pytest.raises(NotImplementedError, match="stg_scale"):
        generate.generate_video("a prompt", **dict(kw, stg_scale=1.0))
    with pytest.raises(NotImplementedError, match="use_fp16=False"):
        generate.generate_video("a prompt", **dict(kw, use_fp16=False))
    with pytest.raises(ValueError, match="8\\*k \\+ 1"):
        generate.generate_video("a prompt", **dict(kw, num_frames=96))
    with pytest.raises(ValueError, match="divisible by 32"):
        generate.generate_video("a prompt", **dict(kw, height=250))


def test_gemm_k_loop_generator_checks_its_own_pipeline():
    """ltx-2-mlx_amd/csrc/gen_gemm_v4.py emits the hand-scheduled K loops AND replays each schedule against the LDS-stage / fragment
    protocol (read only after the DMA landed + a barrier, DMA only after the last read retired + a barrier, no MFMA on an unretired
    fragment).  The shipped variants must pass; schedules that break the protocol must be caught -- that is the point of the checker."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltx-2-mlx_amd", "csrc", "gen_gemm_v4.py")
    spec = importlib.util.spec_from_file_location("gen_gemm_v4", path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    d14 = list(range(0, 56, 4)) + [55]
    ok = gen.Gen(14, 4, mb=16, npa=7, dma_last=d14, dma_ks0=[], m0_early=True)
    ok.build()
    assert gen.check(ok) == []
    short = gen.Gen(6, 4, mb=16, npa=3, dma_last=list(range(1, 23, 2)), dma_ks0=[], m0_early=True)       # the ragged-last-row-tile loop
    short.build()
    assert gen.check(short) == []
    # the checker must notice a broken protocol: replay the SAME schedule with its barriers, its DMA waits or its fragment waits removed
    for drop in ("barrier", "vm", "lgkm0"):
        bad = gen.Gen(14, 4, mb=16, npa=7, dma_last=d14, dma_ks0=[], m0_early=True)
        bad.build()
        bad.trace = [e for e in bad.trace if e[0] != drop]
        assert gen.check(bad), drop
