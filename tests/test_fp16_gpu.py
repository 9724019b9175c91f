"""float16 operand mode (the reference's default compute dtype, scripts/generate.py:1006, 2412-2423): the same kernels compiled with
IEEE-half activations / weights (libltx2hip_f16.so, -DLTX2_F16; v_mfma_f32_16x16x32_f16 / 32x32x16_f16, fp32 accumulation, fp32
residual stream) behind LTXModel(compute_dtype=torch.float16) and SimpleVideoDecoder(compute_dtype=torch.float16), against the fp32
oracle.  Tolerances: float16 carries 11 significant bits against bfloat16's 8, so the bars are tighter than the bf16 ones."""
import math

import pytest
import torch

from conftest import rel_l2
from test_parity import inputs, pearson

pytestmark = pytest.mark.gpu
F16 = torch.float16


def make_dit16(dev, heads, layers, cap, seed=0):
    from oracle import dit
    from ltx_2_mlx_amd.model.transformer import LTXModel
    cfg = dit.DiTConfig(num_attention_heads=heads, attention_head_dim=128, num_layers=layers, caption_channels=cap)
    w = dit.make_dit_weights(cfg, seed)
    wq = {k: (v.to(F16).float() if (k.endswith(".weight") and v.dim() == 2) else v) for k, v in w.items()}     # the oracle sees the f16-rounded weights
    m = LTXModel(num_attention_heads=heads, attention_head_dim=128, num_layers=layers, caption_channels=cap, compute_dtype=F16, device=dev)
    m.load_state_dict(w)
    return cfg, wq, m


def test_leaf_kernels_in_float16(dev):
    """GEMM (every epilogue the DiT uses), fused-QKV V^T, flash attention (plain grid and stream-K) and the norm / RoPE row kernels on
    float16 tensors against fp64 / fp32 torch references of the same ops."""
    import ltx_2_mlx_amd.kernels as K
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator().manual_seed(1)
    M, N, Kk = 3456, 1024, 4096
    a = torch.randn(M, Kk, generator=g).to(F16).to(dev)
    w = (torch.randn(N, Kk, generator=g) / math.sqrt(Kk)).to(F16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ w.double().t() + b.double()
    out = K.gemm(a, w, b)
    assert out.dtype == F16 and rel_l2(out.cpu().float(), ref.cpu()) < 0.0008
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_F32).cpu(), ref.cpu()) < 1e-5
    og = K.gemm(a, w, b, epilogue=nv.EPI_GELU_BF16)
    assert rel_l2(og.cpu().float(), torch.nn.functional.gelu(ref.float(), approximate="tanh").cpu()) < 0.0008
    x = torch.randn(M, N, generator=g).to(dev)
    x2 = x.clone()
    gt = torch.randn(N, generator=g).to(dev)
    K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x2, gate_table=gt)
    assert rel_l2(x2.cpu(), (x.double() + gt.double()[None] * ref).cpu()) < 1e-5
    # attention: 8 heads x 128, ragged N
    H, hd, Nq = 8, 128, 1000
    D = H * hd
    q, k, v = [torch.randn(Nq, D, generator=g).to(F16).to(dev) for _ in range(3)]
    vt = K.vt_transpose(v, H)
    o = K.flash_attn(q, k, vt, H, Nq)
    qh, kh, vh = [t.float().reshape(Nq, H, hd).transpose(0, 1) for t in (q, k, v)]
    ro = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh
    assert o.dtype == F16 and rel_l2(o.cpu().float(), ro.transpose(0, 1).reshape(Nq, D).cpu()) < 0.0012
    q2, k2, v2 = [torch.randn(3456, D, generator=g).to(F16).to(dev) for _ in range(3)]
    vt2 = K.vt_transpose(v2, H)
    o2 = K.flash_attn(q2, k2, vt2, H, 3456)
    assert torch.equal(o2, K.flash_attn(q2, k2, vt2, H, 3456))
    q2h, k2h, v2h = [t.float().reshape(3456, H, hd).transpose(0, 1) for t in (q2, k2, v2)]
    r2 = (torch.softmax(q2h @ k2h.transpose(1, 2) / math.sqrt(hd), dim=-1) @ v2h).transpose(0, 1).reshape(3456, D)
    assert rel_l2(o2.cpu().float(), r2.cpu()) < 2e-3          # a 54-tile row in IEEE half (P <= 2^12 by the kernel's sum check)
    # fused QKV with the V^T epilogue == GEMM + transpose pass
    wq = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(F16).to(dev)
    xin = torch.randn(3456, D, generator=g).to(F16).to(dev)
    out3, vt3, fused = K.gemm_qkv_vt(xin, wq, None, H, hd)
    full = K.gemm(xin, wq, None)
    assert fused and torch.equal(out3[:, :2 * D], full[:, :2 * D]) and torch.equal(vt3, K.vt_transpose(full[:, 2 * D:], H))
    # norm + modulation
    xf = torch.randn(100, 1024, generator=g).to(dev)
    tab = (0.1 * torch.randn(2, 1024, generator=g)).to(dev)
    y = K.adaln_rmsnorm(xf, scale_tab=tab[1], shift_tab=tab[0], dtype=F16)
    ry = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6) * (1 + tab[1]) + tab[0]
    assert y.dtype == F16 and rel_l2(y.float(), ry) < 0.0008


def test_dit_step_and_loop_in_float16(dev):
    """Tiny DiT (2 layers, 2 x 128 heads) in float16: x0 against the fp32 oracle (scalar and per-token timesteps), the fused step, and
    the hipGraph replay of the 8-step loop."""
    from oracle import dit, loop
    from ltx_2_mlx_amd.model.transformer import Modality, X0Model
    cfg, wq, m = make_dit16(dev, heads=2, layers=2, cap=128, seed=3)
    f, h, wd = 3, 6, 8
    lat, ctx, pos = inputs(f, h, wd, 40, 128, seed=4)
    for ts in (torch.tensor([0.725]), (torch.rand(1, f * h * wd, 1, generator=torch.Generator().manual_seed(5)) > 0.3).float() * 0.909375):
        ref = dit.x0_model(lat, ctx, ts, pos, wq, cfg)
        x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
        assert rel_l2(x0.cpu(), ref) < 0.0003 and pearson(x0.cpu(), ref) > 0.9999
    sig = loop.DISTILLED_SIGMA_VALUES
    ref = loop.denoise_loop_cli(loop.unpatchify(lat, f, h, wd), lambda tok, s: dit.x0_model(tok, ctx, torch.tensor([s]), pos, wq, cfg), sig)
    m.prepare(ctx.to(dev), pos.to(dev))
    z = lat[0].to(dev).contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.capture_denoise_graph(z, sig)
        m.replay_denoise_graph()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert rel_l2(z.cpu(), loop.patchify(ref)[0]) < 0.0002


def test_vae_decode_in_float16(dev):
    from oracle import vae
    from ltx_2_mlx_amd.model.video_vae import SimpleVideoDecoder
    blocks = [["res_x", {"num_layers": 1}], ["compress_all", {"multiplier": 2, "residual": True}], ["res_x", {"num_layers": 1}]]
    vcfg = vae.VAEConfig(decoder_blocks=blocks, base_channels=16, timestep_conditioning=True)
    vw = vae.make_vae_weights(vcfg, 1)
    vwq = {k: (v.to(F16).float() if (v.dim() == 5 or (v.dim() == 2 and "linear" in k)) else v) for k, v in vw.items()}
    d = SimpleVideoDecoder(decoder_blocks=blocks, base_channels=16, timestep_conditioning=True, compute_dtype=F16, device=dev)
    d.load_state_dict(vw)
    g = torch.Generator().manual_seed(2)
    z, nz = torch.randn(1, 128, 2, 3, 4, generator=g), torch.randn(1, 128, 2, 3, 4, generator=g)
    ref = vae.decoder_forward(z, vwq, vcfg, 0.05, noise=nz)
    out = d(z.to(dev), timestep=0.05, noise=nz.to(dev)).cpu()
    assert out.shape == ref.shape and rel_l2(out, ref) < 0.0025


def test_dit_48_layer_step_in_float16(dev):
    """The headline geometry (48 layers, D = 4096, N = 3456, S = 1024) in float16 against the fp32 oracle executed on the GPU:
    rel-L2 <= 1e-2."""
    from oracle import dit
    from test_parity_fullsize import dit_weights_on_gpu
    from ltx_2_mlx_amd.model.transformer import LTXModel, Modality, X0Model
    cfg = dit.DiTConfig(num_layers=48)
    w = dit_weights_on_gpu(cfg, dev, seed=48)
    w = {k: (v.to(F16).float() if (k.endswith(".weight") and v.dim() == 2) else v) for k, v in w.items()}
    m = LTXModel(num_layers=48, compute_dtype=F16, device=dev)
    m.load_state_dict(w)
    lat, ctx, pos = inputs(9, 16, 24, 1024, 3840, seed=49)
    for sigma in (1.0, 0.421875):
        ts = torch.tensor([sigma])
        with torch.device(dev), torch.no_grad():
            ref = dit.x0_model(lat.to(dev), ctx.to(dev), ts.to(dev), pos.to(dev), w, cfg).cpu()
        x0 = X0Model(m)(Modality(latent=lat.to(dev), context=ctx.to(dev), context_mask=None, timesteps=ts.to(dev), positions=pos.to(dev)))
        e, r = rel_l2(x0.cpu(), ref), pearson(x0.cpu(), ref)
        print(f"float16, 48 layers, sigma {sigma}: rel-L2 {e:.5f}, Pearson {r:.6f}")
        assert e < 2e-3 and r > 0.9999, (sigma, e, r)       # measured 4.0e-4 (round 5)
    del w, m
    torch.cuda.empty_cache()


def test_text_cross_attention_forms_in_float16(dev):
    """Round 3's two text-cross-attention forms on float16 tensors: the boolean key mask (attention.py:38-70) and q_norm folded into the
    softmax row scale (GEMM epilogue partial sums), against fp64 math."""
    import ltx_2_mlx_amd.kernels as K
    g = torch.Generator().manual_seed(5)
    H, hd, Nq, S = 32, 128, 3456, 1024
    D = H * hd
    q = torch.randn(Nq, D, generator=g).to(F16).to(dev)
    k = torch.randn(S, D, generator=g).to(F16).to(dev)
    v = torch.randn(S, D, generator=g).to(F16).to(dev)
    vt = K.vt_transpose(v, H)
    mk = torch.rand(S, generator=g) > 0.3
    mk[:64] = False                                     # a whole KV tile masked
    out = K.flash_attn_keymask(q, k, vt, H, S, mk)
    qh, kh, vh = [t.double().cpu().reshape(-1, H, hd).transpose(0, 1) for t in (q, k, v)]
    s = (qh @ kh.transpose(1, 2)) / math.sqrt(hd) + (1 - mk.double()) * -3.4e38
    ref = (torch.softmax(s, dim=-1) @ vh).transpose(0, 1).reshape(Nq, D)
    assert out.dtype == F16 and rel_l2(out.double().cpu(), ref) < 0.0012
    # q_norm fold: projection with row partial sums, keys carrying k_norm.weight * q_norm.weight
    x = torch.randn(Nq, D, generator=g).to(F16).to(dev)
    wq = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(F16).to(dev)
    qn, kn = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    qp, rowss = K.gemm_rowss(x, wq, None)
    assert rowss is not None and qp.dtype == F16
    kb = k.clone()
    K.qknorm_rope_(kb, D, hd, 0, kn * qn)
    o2 = K.flash_attn_rowscale(qp, kb, vt, H, S, rowss)
    qf = qp.double().cpu()
    qf = qf * torch.rsqrt((qf * qf).mean(-1, keepdim=True) + 1e-6) * qn.double().cpu()
    kf = k.double().cpu()
    kf = kf * torch.rsqrt((kf * kf).mean(-1, keepdim=True) + 1e-6) * kn.double().cpu()
    qh2, kh2 = [t.reshape(-1, H, hd).transpose(0, 1) for t in (qf, kf)]
    ref2 = (torch.softmax(qh2 @ kh2.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh).transpose(0, 1).reshape(Nq, D)
    assert o2.dtype == F16 and rel_l2(o2.double().cpu(), ref2) < 0.0015
