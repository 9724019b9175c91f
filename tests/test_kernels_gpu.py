"""Per-kernel parity: HIP kernel (through the C ABI) vs the CPU oracle / plain fp32 torch on the
same seeded inputs.  Tolerances are for bf16 operands with fp32 accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def q(t):
    """Round to bf16 and back, so the fp32 reference sees exactly the kernel's operands."""
    return t.to(BF).float()


@pytest.fixture(scope="module")
def K(dev):
    import ltx_2_mlx_amd.kernels as k
    return k


@pytest.mark.parametrize("M,N,Kd", [(128, 128, 64), (256, 384, 256), (288, 768, 1024), (3456, 4096, 128), (100, 128, 192), (37, 48, 128)])
def test_gemm_bias(K, dev, M, N, Kd):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = q(torch.randn(M, Kd, generator=g))
    w = q(torch.randn(N, Kd, generator=g) / math.sqrt(Kd))
    b = torch.randn(N, generator=g)
    ref = a @ w.t() + b
    out = K.gemm(a.to(dev, BF), w.to(dev, BF), b.to(dev), epilogue=3)   # fp32 out
    assert rel_l2(out.cpu(), ref) < 1e-05
    # asymmetric-operand transpose check: distinct row/col patterns
    out_bf = K.gemm(a.to(dev, BF), w.to(dev, BF), b.to(dev), epilogue=0)
    assert rel_l2(out_bf.float().cpu(), ref) < 6e-3


def test_gemm_epilogues(K, dev):
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator().manual_seed(5)
    M, N, Kd = 200, 256, 320
    a = q(torch.randn(M, Kd, generator=g))
    w = q(torch.randn(N, Kd, generator=g) / math.sqrt(Kd))
    b = torch.randn(N, generator=g)
    lin = a @ w.t() + b
    A, W, B = a.to(dev, BF), w.to(dev, BF), b.to(dev)
    assert rel_l2(K.gemm(A, W, B, epilogue=nv.EPI_GELU_BF16).float().cpu(), F.gelu(lin, approximate="tanh")) < 8e-3
    assert rel_l2(K.gemm(A, W, B, epilogue=nv.EPI_SILU_BF16).float().cpu(), F.silu(lin)) < 8e-3
    # residual + gate accumulate: per-row gate + table, broadcast gate, no gate
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(M, N, generator=g)
    tab = torch.randn(N, generator=g)
    x = x0.clone().to(dev)
    K.gemm(A, W, B, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate=gate.to(dev), gate_table=tab.to(dev))
    assert rel_l2(x.cpu(), x0 + (gate + tab) * lin) < 1e-05
    x = x0.clone().to(dev)
    K.gemm(A, W, B, epilogue=nv.EPI_RESID_GATE_F32, out=x, gate=gate[:1].contiguous().to(dev), gate_table=tab.to(dev))
    assert rel_l2(x.cpu(), x0 + (gate[:1] + tab) * lin) < 1e-05
    x = x0.clone().to(dev)
    K.gemm(A, W, B, epilogue=nv.EPI_RESID_GATE_F32, out=x)
    assert rel_l2(x.cpu(), x0 + lin) < 1e-05
    res = q(torch.randn(M, N, generator=g))
    o = K.gemm(A, W, B, epilogue=nv.EPI_ADD_BF16, res=res.to(dev, BF))
    assert rel_l2(o.float().cpu(), lin + res) < 8e-3


@pytest.mark.parametrize("M", [3456, 1216, 1217, 1250, 1280, 1300, 1344])
def test_gemm_v4_ragged_last_row_tile(K, dev, M):
    """The 4-wave asm-loop kernel runs a SHORT K loop on the ragged last row tile of a 224-row grid (6 or 10 of 14 row blocks:
    M mod 224 <= 96 / <= 160) -- every remainder class against an fp32 reference, every epilogue the DiT uses, bf16 and fp8-resident
    weights (bit-identical to each other), and no write outside the M rows."""
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator(device=dev).manual_seed(M)
    N, Kd = 8192, 512
    a = torch.randn(M, Kd, generator=g, device=dev).to(BF)
    w32 = torch.randn(N, Kd, generator=g, device=dev) / math.sqrt(Kd)
    scale = torch.full((N,), float(w32.abs().max() / 448.0), device=dev)
    codes = (w32 / scale[:, None]).to(torch.float8_e4m3fn)
    w = (codes.float() * scale[:, None]).to(BF)
    b = torch.randn(N, generator=g, device=dev)
    lin = a.float() @ w.float().t() + b
    canary = torch.full((M + 300, N), 7.0, device=dev, dtype=BF)
    out = K.gemm(a, w, b, out=canary[:M])
    assert rel_l2(out.float(), lin) < 6e-3 and bool((canary[M:] == 7.0).all())
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_F32), lin) < 1e-05
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_GELU_BF16).float(), F.gelu(lin, approximate="tanh")) < 8e-3
    gate = torch.randn(N, generator=g, device=dev)
    x0 = torch.randn(M + 300, N, generator=g, device=dev)
    x = x0.clone()
    K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x[:M], gate_table=gate)
    assert rel_l2(x[:M], x0[:M] + gate * lin) < 1e-05 and torch.equal(x[M:], x0[M:])
    assert torch.equal(K.gemm_w8a16(a, codes.view(torch.uint8), scale, b), out)
    x8 = x0.clone()
    K.gemm_w8a16(a, codes.view(torch.uint8), scale, b, epilogue=nv.EPI_RESID_GATE_F32, out=x8[:M], gate_table=gate)
    assert torch.equal(x8, x)


@pytest.mark.parametrize("M,N,Kd", [(68, 2048, 2048), (68, 6144, 2048), (68, 8192, 2048), (68, 2048, 8192), (68, 2048, 4096), (1, 256, 256), (16, 48, 512),
                                    (128, 4096, 2048), (37, 8192, 256), (129, 2048, 2048)])
def test_gemm_skinny_m(K, dev, M, N, Kd):
    """M <= 128 rows (the 68-token audio stream of the AudioVideo DiT): the weight-streaming kernel (N cut into 16/32-column strips,
    K cut over the 4 waves) against an fp32 reference for every dense epilogue, no write outside the M rows, bit-reproducible.
    (129 rows: one past its limit -- the tile kernel takes it.)"""
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator(device=dev).manual_seed(M + N + Kd)
    a = torch.randn(M, Kd, generator=g, device=dev).to(BF)
    w = (torch.randn(N, Kd, generator=g, device=dev) / math.sqrt(Kd)).to(BF)
    b = torch.randn(N, generator=g, device=dev)
    lin = a.float() @ w.float().t() + b
    canary = torch.full((M + 40, N), 7.0, device=dev, dtype=BF)
    out = K.gemm(a, w, b, out=canary[:M])
    assert rel_l2(out.float(), lin) < 6e-3 and bool((canary[M:] == 7.0).all())
    for _ in range(3):
        assert torch.equal(K.gemm(a, w, b), out)
    assert rel_l2(K.gemm(a, w, None, epilogue=nv.EPI_F32), lin - b) < 1e-05
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_GELU_BF16).float(), F.gelu(lin, approximate="tanh")) < 8e-3
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_SILU_BF16).float(), F.silu(lin)) < 8e-3
    gate = torch.randn(M, N, generator=g, device=dev)
    tab = torch.randn(N, generator=g, device=dev)
    x0 = torch.randn(M + 40, N, generator=g, device=dev)
    x = x0.clone()
    K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x[:M], gate=gate, gate_table=tab)
    assert rel_l2(x[:M], x0[:M] + (gate + tab) * lin) < 1e-05 and torch.equal(x[M:], x0[M:])
    x = x0.clone()
    K.gemm(a, w, b, epilogue=nv.EPI_RESID_GATE_F32, out=x[:M], gate=gate[:1].contiguous(), gate_table=tab)
    assert rel_l2(x[:M], x0[:M] + (gate[:1] + tab) * lin) < 1e-05
    res = torch.randn(M, N, generator=g, device=dev).to(BF)
    assert rel_l2(K.gemm(a, w, b, epilogue=nv.EPI_ADD_BF16, res=res).float(), lin + res.float()) < 8e-3
    # fp8-resident weights on the same kernel: bit-identical to the bf16 run on the dequantised weights
    if M <= 128:
        w32 = torch.randn(N, Kd, generator=g, device=dev) / math.sqrt(Kd)
        scale = torch.full((N,), float(w32.abs().max() / 448.0), device=dev)
        codes = (w32 / scale[:, None]).to(torch.float8_e4m3fn)
        wdq = K.dequant_fp8(codes.view(torch.uint8), float(scale[0]))
        assert torch.equal(K.gemm_w8a16(a, codes.view(torch.uint8), scale, b), K.gemm(a, wdq, b))
        xa, xb = x0.clone(), x0.clone()
        K.gemm_w8a16(a, codes.view(torch.uint8), scale, b, epilogue=nv.EPI_RESID_GATE_F32, out=xa[:M], gate_table=tab)
        K.gemm(a, wdq, b, epilogue=nv.EPI_RESID_GATE_F32, out=xb[:M], gate_table=tab)
        assert torch.equal(xa, xb)
    # strided activations (a column slice of a wider buffer, as the engine passes them)
    wide = torch.randn(M, Kd + 64, generator=g, device=dev).to(BF)
    assert rel_l2(K.gemm(wide[:, 64:], w, b).float(), wide[:, 64:].float() @ w.float().t() + b) < 6e-3


def test_gemm_rejects_bad_k(K, dev):
    a = torch.zeros(8, 72, device=dev, dtype=BF)
    w = torch.zeros(128, 72, device=dev, dtype=BF)
    with pytest.raises(ValueError):
        K.gemm(a, w)


@pytest.mark.parametrize("M", [1, 3, 9])
def test_gemv(K, dev, M):
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, 512, generator=g)
    w = q(torch.randn(300, 512, generator=g) / 20)
    b = torch.randn(300, generator=g)
    ref = F.silu(F.silu(a) @ w.t() + b)
    out = K.gemv(a.to(dev), w.to(dev, BF), b.to(dev), act_in=1, act_out=1)
    assert rel_l2(out.cpu(), ref) < 1e-05


@pytest.mark.parametrize("rows,D", [(5, 256), (288, 4096)])
def test_adaln_rmsnorm(K, dev, rows, D):
    from oracle import dit
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 3
    tab = 0.1 * torch.randn(6, D, generator=g)
    emb = 0.1 * torch.randn(rows, 6, D, generator=g)
    # per-token modulation
    ref = dit.adaln_forward(x, tab[1] + emb[:, 1], tab[0] + emb[:, 0], 1e-6)
    e = emb.to(dev).contiguous()
    t = tab.to(dev).contiguous()
    out = K.adaln_rmsnorm(x.to(dev), scale_tab=t[1], shift_tab=t[0], scale_emb=e[:, 1], shift_emb=e[:, 0], emb_stride=6 * D)
    assert rel_l2(out.float().cpu(), ref) < 4e-3
    # broadcast modulation (stride 0) and plain rms
    ref = dit.adaln_forward(x, tab[1] + emb[0, 1], tab[0] + emb[0, 0], 1e-6)
    out = K.adaln_rmsnorm(x.to(dev), scale_tab=t[1], shift_tab=t[0], scale_emb=e[0, 1], shift_emb=e[0, 0], emb_stride=0)
    assert rel_l2(out.float().cpu(), ref) < 4e-3
    assert rel_l2(K.adaln_rmsnorm(x.to(dev)).float().cpu(), dit.rms_norm(x)) < 4e-3
    # LayerNorm (output head)
    ref = F.layer_norm(x, (D,), eps=1e-6) * (1 + tab[1]) + tab[0]
    out = K.adaln_rmsnorm(x.to(dev), layer_norm=True, scale_tab=t[1], shift_tab=t[0])
    assert rel_l2(out.float().cpu(), ref) < 4e-3


@pytest.mark.parametrize("heads,f,h,w", [(2, 3, 4, 4), (32, 2, 3, 5)])
def test_qknorm_rope(K, dev, heads, f, h, w):
    from oracle import dit, loop
    D = heads * 128
    N = f * h * w
    g = torch.Generator().manual_seed(heads)
    qkv = q(torch.randn(N, 3 * D, generator=g))
    wq = 1 + 0.1 * torch.randn(D, generator=g)
    wk = 1 + 0.1 * torch.randn(D, generator=g)
    pos = loop.video_positions(1, f, h, w, 24.0)
    cos, sin = dit.rope_split_tables(pos, D, heads, 10000.0, [20, 2048, 2048])
    rq = dit.apply_split_rope(dit.rms_norm(qkv[None, :, :D], wq), cos, sin)[0]
    rk = dit.apply_split_rope(dit.rms_norm(qkv[None, :, D:2 * D], wk), cos, sin)[0]
    # token-major tables [N, D/2]: slot h*64 + j
    cos_t = cos[0].permute(1, 0, 2).reshape(N, D // 2).contiguous().to(dev)
    sin_t = sin[0].permute(1, 0, 2).reshape(N, D // 2).contiguous().to(dev)
    buf = qkv.to(dev, BF).contiguous()
    K.qknorm_rope_(buf, D, 128, 0, wq.to(dev), D, wk.to(dev), 1e-6, cos_t, sin_t)
    assert rel_l2(buf[:, :D].float().cpu(), rq) < 6e-3
    assert rel_l2(buf[:, D:2 * D].float().cpu(), rk) < 6e-3
    assert torch.equal(buf[:, 2 * D:].cpu(), qkv[:, 2 * D:].to(BF))        # v untouched
    # no-rope, single segment
    buf = qkv[:, :D].to(dev, BF).contiguous()
    K.qknorm_rope_(buf, D, 128, 0, wq.to(dev))
    assert rel_l2(buf.float().cpu(), dit.rms_norm(qkv[:, :D], wq)) < 6e-3


@pytest.mark.parametrize("heads,Nq,Nkv", [(2, 128, 64), (2, 288, 288), (3, 100, 37), (32, 288, 1024), (2, 130, 3456)])
def test_flash_attn(K, dev, heads, Nq, Nkv):
    from oracle import dit
    D = heads * 128
    g = torch.Generator().manual_seed(Nq + Nkv)
    qq = q(torch.randn(Nq, D, generator=g))
    kk = q(torch.randn(Nkv, D, generator=g))
    vv = q(torch.randn(Nkv, D, generator=g))
    ref = dit.sdpa(qq[None], kk[None], vv[None], heads)[0]
    vt = K.vt_transpose(vv.to(dev, BF), heads)
    out = K.flash_attn(qq.to(dev, BF), kk.to(dev, BF), vt, heads, Nkv)
    assert rel_l2(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("heads,Nq,Nkv", [(2, 130, 37), (4, 37, 700), (32, 68, 68), (1, 256, 64)])
def test_flash_attn_head_dim_64(K, dev, heads, Nq, Nkv):
    """Audio-stream / audio<->video attention geometry: head_dim 64, ragged and tiny token counts."""
    from oracle import dit
    D = heads * 64
    g = torch.Generator().manual_seed(Nq * 7 + Nkv)
    qq = q(torch.randn(Nq, D, generator=g))
    kk = q(torch.randn(Nkv, D, generator=g))
    vv = q(torch.randn(Nkv, D, generator=g))
    ref = dit.sdpa(qq[None], kk[None], vv[None], heads)[0]
    vt = K.vt_transpose(vv.to(dev, BF), heads, head_dim=64)
    assert vt.shape == (heads, 64, (Nkv + 63) // 64 * 64)
    out = K.flash_attn(qq.to(dev, BF), kk.to(dev, BF), vt, heads, Nkv)
    assert out.shape == (Nq, D)
    assert rel_l2(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("rows,Dq,heads,hd", [(3456 // 8, 4096, 32, 128), (68, 2048, 32, 64), (37, 512, 4, 64), (50, 4096, 32, 64)])
def test_attn_head_gate(K, dev, rows, Dq, heads, hd):
    """to_gate_logits + 2*sigmoid gating (reference attention.py:241-249)."""
    g = torch.Generator().manual_seed(rows + Dq)
    x = q(torch.randn(rows, Dq, generator=g))
    wg = q(torch.randn(heads, Dq, generator=g) / math.sqrt(Dq))
    bg = torch.randn(heads, generator=g)
    att = q(torch.randn(rows, heads * hd, generator=g))
    logits_ref = x @ wg.t() + bg
    ref = (att.reshape(rows, heads, hd) * (2 * torch.sigmoid(logits_ref))[..., None]).reshape(rows, -1)
    a = att.to(dev, BF).contiguous()
    logits = K.attn_head_gate_(a, x.to(dev, BF), wg.to(dev, BF), bg.to(dev), heads)
    assert (logits.cpu() - logits_ref).abs().max() < 2e-3
    assert rel_l2(a.float().cpu(), ref) < 6e-3


def test_flash_attn_forced_rescale(K, dev):
    """Spike one key against one query at a late tile so the running max jumps mid-stream."""
    from oracle import dit
    heads, Nq, Nkv = 1, 64, 512
    g = torch.Generator().manual_seed(0)
    qq = q(torch.randn(Nq, 128, generator=g))
    kk = q(torch.randn(Nkv, 128, generator=g))
    vv = q(torch.randn(Nkv, 128, generator=g))
    kk[300] = qq[7] * 4.0
    kk[301] = qq[33] * 6.0
    kk = q(kk)
    ref = dit.sdpa(qq[None], kk[None], vv[None], heads)[0]
    vt = K.vt_transpose(vv.to(dev, BF), heads)
    out = K.flash_attn(qq.to(dev, BF), kk.to(dev, BF), vt, heads, Nkv)
    assert rel_l2(out.float().cpu(), ref) < 0.006
    assert (out.float().cpu() - ref).abs().max() < 5e-2


@pytest.mark.parametrize("heads,Nq,Nkv,hd", [(32, 3456, 3456, 128), (32, 3456, 1024, 128), (32, 3456, 68, 64), (4, 300, 1000, 128)])
def test_flash_attn_bit_reproducible(K, dev, heads, Nq, Nkv, hd):
    """No atomics anywhere: repeated launches on the same operands must agree bit for bit.  At full grid size this
    is what catches a hand-scheduled instruction that reads an MFMA result before it has landed (the row-max
    chain is asm) -- the values stay within tolerance, the bits do not."""
    g = torch.Generator().manual_seed(5)
    D = heads * hd
    qq = torch.randn(Nq, D, generator=g).to(dev, BF)
    kk = torch.randn(Nkv, D, generator=g).to(dev, BF)
    vt = K.vt_transpose(torch.randn(Nkv, D, generator=g).to(dev, BF), heads, head_dim=hd)
    ref = K.flash_attn(qq, kk, vt, heads, Nkv).clone()
    for _ in range(8):
        assert torch.equal(K.flash_attn(qq, kk, vt, heads, Nkv), ref)


@pytest.mark.parametrize("heads,Nq,Nkv,hd", [(32, 3456, 3456, 128), (32, 3456, 1024, 128), (32, 3456, 68, 64), (32, 3400, 3401, 128),
                                              (20, 4000, 999, 128), (32, 13824, 1024, 128), (8, 8300, 130, 64),
                                              (32, 68, 3456, 64), (2, 100, 5000, 128), (3, 37, 1100, 64)])
def test_flash_attn_grid_shapes(K, dev, heads, Nq, Nkv, hd):
    """Grids of one round, several rounds and a partly filled last round of workgroup slots; ragged query / key counts; few queries against a long
    key range: against the oracle, bit-reproducible."""
    from oracle import dit
    g = torch.Generator().manual_seed(Nq + Nkv + hd)
    D = heads * hd
    q32, k32, v32 = (q(torch.randn(n, D, generator=g)) for n in (Nq, Nkv, Nkv))
    qq, kk = q32.to(dev, BF), k32.to(dev, BF)
    vt = K.vt_transpose(v32.to(dev, BF), heads, head_dim=hd)
    out = K.flash_attn(qq, kk, vt, heads, Nkv).clone()
    for _ in range(6):
        assert torch.equal(K.flash_attn(qq, kk, vt, heads, Nkv), out)
    if Nq * Nkv <= 3456 * 3456:
        ref = dit.sdpa(q32[None], k32[None], v32[None], heads)[0]
        assert rel_l2(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("case", ["spike", "overflow", "ramp", "first_tile_low"])
def test_flash_attn_stale_maximum_paths(K, dev, case):
    """The kernel keeps the FIRST tile's row maximum as the softmax reference and only re-references a row when a tile's exponentials sum past 2^30
    (attention.hip: stale maximum).  Scores that climb after the first tile must take that classic path and come out right:
    spike = one key 65 / 100 exp2-units above the first tile's maximum (finite exponentials, the sum check fires); overflow = a key ~200 units above
    (exp2 overflows to inf: caught by the same check, the scores are still intact); ramp = key norms growing x15 along the sequence (several
    re-references per row); first_tile_low = the first tile's keys scaled to ~0 (the reference starts far BELOW the row's maximum, P grows to 2^20:
    no re-reference, fp32 accumulation carries it)."""
    heads, hd, Nq, Nkv = 2, 128, 200, 1000
    g = torch.Generator().manual_seed(11)
    D = heads * hd
    qq = torch.randn(Nq, D, generator=g)
    kk = torch.randn(Nkv, D, generator=g)
    vv = torch.randn(Nkv, D, generator=g)
    if case == "spike":
        kk[300, :hd] = qq[7, :hd] * 4.0
        kk[700, hd:] = qq[133, hd:] * 6.0
    elif case == "overflow":
        kk[500, :hd] = qq[100, :hd] * 12.0
        kk[130, hd:] = qq[3, hd:] * 14.0
    elif case == "ramp":
        kk = kk * torch.linspace(0.2, 3.0, Nkv)[:, None]
    else:
        kk[:64] *= 0.01
        kk[64:] *= 4.0
    qq, kk, vv = qq.to(BF), kk.to(BF), vv.to(BF)
    qh, kh, vh = [t.double().reshape(-1, heads, hd).transpose(0, 1) for t in (qq, kk, vv)]
    exact = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh).transpose(0, 1).reshape(Nq, D)
    vt = K.vt_transpose(vv.to(dev), heads, head_dim=hd)
    out = K.flash_attn(qq.to(dev), kk.to(dev), vt, heads, Nkv)
    assert torch.isfinite(out).all()
    assert rel_l2(out.double().cpu(), exact) < 1e-2
    assert torch.equal(out, K.flash_attn(qq.to(dev), kk.to(dev), vt, heads, Nkv))


def test_flash_attn_beside_a_busy_stream(K, dev):
    """Attention launches while a second stream keeps GEMMs and row kernels in flight on the same GPU -- the AudioVideo engine's situation: every
    launch bit-identical (what catches a hand-issued LDS read or MFMA result consumed before it has landed)."""
    N, D, H = 3456, 4096, 32
    g = torch.Generator(device=dev).manual_seed(0)
    qq, kk = (torch.randn(N, D, generator=g, device=dev).to(BF) for _ in range(2))
    vt = K.vt_transpose(torch.randn(N, D, generator=g, device=dev).to(BF), H)
    ref = K.flash_attn(qq, kk, vt, H, N).clone()
    side = torch.cuda.Stream()
    a = torch.randn(3456, 4096, device=dev).to(BF)
    w = (torch.randn(4096, 4096, device=dev) / 64).to(BF)
    x = torch.randn(3456, 4096, device=dev)
    for i in range(300):
        with torch.cuda.stream(side):
            if i % 3 == 0:
                K.gemm(a, w)
            elif i % 3 == 1:
                K.adaln_rmsnorm(x)
            else:
                x.mul_(1.0)
        out = K.flash_attn(qq, kk, vt, H, N)
        if i % 25 == 0:
            assert torch.equal(out, ref), i
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,heads,hd,Kd,expect_fused", [(3456, 32, 128, 4096, True), (3400, 8, 128, 512, True), (2200, 32, 64, 1024, True), (1100, 16, 64, 1024, False),
                                                       (3360, 8, 128, 512, False), (68, 32, 64, 2048, False)])
def test_gemm_qkv_writes_vt(K, dev, M, heads, hd, Kd, expect_fused):
    """V^T written by the QKV GEMM's epilogue (round 2): bit-identical to GEMM + vt_transpose, including the zero padding of the
    last 64-key block; Q and K columns untouched by the fusion.  3360 = 15 x 224 rows: the row tiles stop short of Npad = 3392, so
    the launcher must fall back to the transpose pass (as for grids the small-tile kernel takes: 1100 x 3072, the 68-token audio stream)."""
    g = torch.Generator().manual_seed(M + hd)
    D = heads * hd
    a = torch.randn(M, Kd, generator=g).to(dev, BF)
    w = (torch.randn(3 * D, Kd, generator=g) / math.sqrt(Kd)).to(dev, BF)
    b = torch.randn(3 * D, generator=g).to(dev)
    ref = K.gemm(a, w, b)
    vt_ref = K.vt_transpose(ref[:, 2 * D:], heads, head_dim=hd)
    out, vt, fused = K.gemm_qkv_vt(a, w, b, heads, hd)
    assert fused == expect_fused
    assert torch.equal(out[:, :2 * D], ref[:, :2 * D])
    assert vt.shape == vt_ref.shape and torch.equal(vt, vt_ref)
    for _ in range(3):
        assert torch.equal(K.gemm_qkv_vt(a, w, b, heads, hd)[1], vt_ref)


@pytest.mark.parametrize("M,N,Kd", [(1024, 8192, 3840), (3456, 4096, 4096), (300, 512, 256)])
def test_gemm_bit_reproducible(K, dev, M, N, Kd):
    """Same for the GEMM tile kernels (128x128 tile with hand-issued fragment reads, 224/256-row ping-pong)."""
    g = torch.Generator().manual_seed(6)
    a = torch.randn(M, Kd, generator=g).to(dev, BF)
    w = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev, BF)
    b = torch.randn(N, generator=g).to(dev)
    ref = K.gemm(a, w, b).clone()
    for _ in range(8):
        assert torch.equal(K.gemm(a, w, b), ref)


def test_flash_attn_strided_views(K, dev):
    """q/k/v as column slices of one fused qkv buffer (how the engine calls it)."""
    from oracle import dit
    heads, N = 2, 200
    D = heads * 128
    g = torch.Generator().manual_seed(3)
    qkv = q(torch.randn(N, 3 * D, generator=g))
    ref = dit.sdpa(qkv[None, :, :D], qkv[None, :, D:2 * D], qkv[None, :, 2 * D:], heads)[0]
    buf = qkv.to(dev, BF)
    vt = K.vt_transpose(buf[:, 2 * D:], heads)
    out = K.flash_attn(buf[:, :D], buf[:, D:2 * D], vt, heads, N)
    assert rel_l2(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("T,H,W,Cin,Cout", [(3, 4, 6, 64, 128), (2, 5, 3, 128, 64), (4, 8, 8, 64, 48), (1, 2, 2, 64, 256)])
def test_conv3d(K, dev, causal, T, H, W, Cin, Cout):
    from oracle import vae
    g = torch.Generator().manual_seed(T * 100 + Cin)
    x = q(torch.randn(1, Cin, T, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = torch.randn(Cout, generator=g)
    ref = vae.conv3d_simple(x, w, b, causal=causal)[0].permute(1, 2, 3, 0)      # T,H,W,C
    xe = x[0].permute(1, 2, 3, 0).contiguous().to(dev, BF)
    out = K.conv3d(xe, K.conv_weight_to_engine(w).to(dev), b.to(dev), causal=causal)
    assert rel_l2(out.float().cpu(), ref) < 6e-3
    res = q(torch.randn(T, H, W, Cout, generator=g))
    out = K.conv3d(xe, K.conv_weight_to_engine(w).to(dev), b.to(dev), causal=causal, mode=1, res=res.to(dev, BF))
    assert rel_l2(out.float().cpu(), ref + res) < 6e-3


@pytest.mark.parametrize("T,H,W,C", [(49, 128, 192, 128), (25, 64, 96, 256), (13, 32, 48, 512), (7, 16, 24, 1024)])
def test_conv3d_decoder_stage_sizes(K, dev, T, H, W, C):
    """The four res-block conv shapes of one 7-latent-frame chunk at 768x512 (BASELINE config 2): up to 1.2 M output rows,
    the 128^2 tile kernel at C = 128 and the 256-wide ping-pong conv kernel above.  Reference: plain fp32 torch on the GPU,
    27 shifted matmuls over the reflect(H, W) / replicate(T) padded volume (simple_decoder.py:146-175)."""
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.randn(T, H, W, C, generator=g, device=dev).to(BF)
    w = (torch.randn(C, C, 3, 3, 3, generator=g, device=dev) / math.sqrt(27 * C)).to(BF)
    b = torch.randn(C, generator=g, device=dev)
    out = K.conv3d(x, K.conv_weight_to_engine(w), b)
    xp = x.float().permute(3, 0, 1, 2)[None]                                       # 1,C,T,H,W
    xp = F.pad(xp.reshape(1, C * T, H, W), (1, 1, 1, 1), mode="reflect").reshape(1, C, T, H + 2, W + 2)
    xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], dim=2)[0].permute(1, 2, 3, 0)   # T+2,H+2,W+2,C
    ref = b.float().expand(T * H * W, C).clone()
    wf = w.float()
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                ref += xp[kt:kt + T, kh:kh + H, kw:kw + W].reshape(-1, C) @ wf[:, :, kt, kh, kw].t()
    ref = ref.reshape(T, H, W, C)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 6e-3
    for sl in ((0, 0, 0), (T - 1, H - 1, W - 1), (T // 2, 0, W - 1)):               # corners / edges: the padding rules
        assert rel_l2(out[sl].float().cpu(), ref[sl].cpu()) < 0.008


@pytest.mark.parametrize("stride,mult,residual", [((2, 2, 2), 2, True), ((2, 2, 2), 1, False), ((1, 2, 2), 2, True), ((2, 1, 1), 2, True)])
def test_conv3d_depth_to_space(K, dev, stride, mult, residual):
    from oracle import vae
    Cin, T, H, W = 128, 3, 4, 5
    sp = stride[0] * stride[1] * stride[2]
    Cout = sp * Cin // mult
    g = torch.Generator().manual_seed(sp + mult)
    x = q(torch.randn(1, Cin, T, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = torch.randn(Cout, generator=g)
    wd = {"u.conv.conv.weight": w, "u.conv.conv.bias": b}
    ref = vae.upsample_block(x, wd, "u", stride, mult, residual, causal=False)[0].permute(1, 2, 3, 0)
    xe = x[0].permute(1, 2, 3, 0).contiguous().to(dev, BF)
    out = K.conv3d(xe, K.conv_weight_to_engine(w, stride).to(dev), K.conv_bias_to_engine(b, stride).to(dev), mode=2,
                   stride=stride, residual=residual)
    assert out.shape == ref.shape
    assert rel_l2(out.float().cpu(), ref) < 6e-3


@pytest.mark.parametrize("C", [64, 128, 256, 512, 1024])
def test_pixnorm_mod_silu(K, dev, C):
    from oracle import vae
    g = torch.Generator().manual_seed(C)
    x = q(torch.randn(1, C, 2, 3, 5, generator=g) * 2)
    tab = 0.2 * torch.randn(4, C, generator=g)
    te = 0.2 * torch.randn(4, C, generator=g)
    ss = tab + te
    ref = F.silu(vae.pixel_norm(x) * (1 + ss[3])[None, :, None, None, None] + ss[2][None, :, None, None, None])
    ref = ref[0].permute(1, 2, 3, 0)
    xe = x[0].permute(1, 2, 3, 0).contiguous().to(dev, BF)
    out = K.pixnorm_mod_silu(xe, tab.to(dev), te.to(dev).reshape(-1), 2, 3)
    assert rel_l2(out.float().cpu(), ref) < 6e-3


@pytest.mark.parametrize("C,P", [(64, 300000), (128, 140000), (1024, 9000)])
def test_pixnorm_mod_silu_grid_stride(K, dev, C, P):
    """More positions than one pass of the capped grid covers (8192 blocks x 256 threads / (C/8 or C/16) lanes per position)."""
    g = torch.Generator(device=dev).manual_seed(C)
    x = (torch.randn(P, C, generator=g, device=dev) * 2).to(BF)
    tab = 0.2 * torch.randn(4, C, generator=g, device=dev)
    xf = x.float()
    ref = F.silu(xf * torch.rsqrt(xf.pow(2).mean(dim=1, keepdim=True) + 1e-6) * (1 + tab[1]) + tab[0])
    out = K.pixnorm_mod_silu(x.reshape(1, 1, P, C), tab, None, 0, 1).reshape(P, C)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 6e-3
    assert rel_l2(out[-7:].float().cpu(), ref[-7:].cpu()) < 6e-3


def test_euler_and_x0(K, dev):
    from oracle import loop
    g = torch.Generator().manual_seed(1)
    x = torch.randn(50, 128, generator=g)
    v = torch.randn(50, 128, generator=g)
    ts = torch.rand(50, generator=g)
    x0 = K.x0_from_velocity(x.to(dev), v.to(dev), ts.to(dev))
    assert torch.allclose(x0.cpu(), x - ts[:, None] * v, atol=1e-6)
    x0s = K.x0_from_velocity(x.to(dev), v.to(dev), torch.tensor([0.7], device=dev))
    assert torch.allclose(x0s.cpu(), x - 0.7 * v, atol=1e-6)
    mask = (torch.rand(50, generator=g) > 0.5).float()
    clean = torch.randn(50, 128, generator=g)
    ref = loop.euler_step(x, loop.post_process_latent(x0.cpu(), mask[:, None], clean), 0.9, 0.7)
    out = K.euler_step(x.to(dev), x0, 0.9, 0.7, mask.to(dev), clean.to(dev))
    assert torch.allclose(out.cpu(), ref, atol=1e-5)
    with pytest.raises(ValueError, match="Sigma can't be 0.0"):
        K.euler_step(x.to(dev), x0, 0.0, 0.0)


def test_timestep_sinusoid(K, dev):
    from oracle import dit
    t = torch.tensor([0.0, 0.421875, 1.0])
    ref = dit.sinusoidal_timestep_embedding(t * 1000.0)
    out = K.timestep_sinusoid(t.to(dev), 1000.0)
    assert torch.allclose(out.cpu(), ref, atol=2e-3)     # fp32 sin/cos of arguments up to 1000 rad


def test_video_to_uint8(K, dev):
    from oracle import vae
    g = torch.Generator().manual_seed(2)
    v = torch.randn(1, 3, 4, 6, 10, generator=g) * 0.8
    ref = vae.to_uint8_frames(v)
    out = K.video_to_uint8(v[0].to(dev))
    d = (out.cpu().int() - ref.int()).abs()
    assert d.max() <= 1 and (d > 0).float().mean() < 1e-3      # truncation at exact .0 boundaries only


def test_dequant_fp8_all_codes(K, dev):
    """Every e4m3fn code x scale against torch's own float8_e4m3fn -> f32 conversion (bit-exact before the
    bf16 rounding; reference loader/fp8_loader.py:14-51)."""
    codes = torch.arange(256, dtype=torch.uint8).repeat(5)[:1275]          # ragged tail (not a multiple of 8)
    ref32 = codes.view(torch.float8_e4m3fn).float()
    for scale in (1.0, 0.0123, 3.5):
        out = K.dequant_fp8(codes.to(dev), scale).float().cpu()
        ref = (ref32 * scale).to(BF).float()
        nan = torch.isnan(ref)
        assert torch.equal(torch.isnan(out), nan)
        assert torch.equal(out[~nan], ref[~nan])


@pytest.mark.parametrize("T,H,W,Cin,Cout", [(3, 5, 6, 64, 128), (1, 4, 4, 128, 64), (2, 9, 7, 64, 64)])
def test_conv3d_zero_pad(K, dev, T, H, W, Cin, Cout):
    """Spatial upscaler's conv3d: zero padding in T/H/W (reference upscaler/spatial.py:20-87)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(T * 10 + H)
    x = q(torch.randn(1, Cin, T, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = torch.randn(Cout, generator=g)
    ref = F.conv3d(x, w, b, padding=1)[0].permute(1, 2, 3, 0)
    xe = x[0].permute(1, 2, 3, 0).contiguous().to(dev, BF)
    out = K.conv3d(xe, K.conv_weight_to_engine(w).to(dev), b.to(dev), pad_zero=True)
    assert rel_l2(out.float().cpu(), ref) < 6e-3


def test_conv2d_pixel_shuffle(K, dev):
    """SpatialRationalResampler: per-frame conv2d (C -> 4C, zero pad) + PixelShuffle(2) (upscaler/spatial.py:184-323)."""
    import torch.nn.functional as F
    T, H, W, C = 2, 5, 6, 64
    g = torch.Generator().manual_seed(9)
    x = q(torch.randn(T, C, H, W, generator=g))
    w = q(torch.randn(4 * C, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = torch.randn(4 * C, generator=g)
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2).permute(0, 2, 3, 1)          # T, 2H, 2W, C
    xe = x.permute(0, 2, 3, 1).contiguous().to(dev, BF)
    out = K.conv3d(xe, K.conv2d_weight_to_engine(w, pixel_shuffle=2).to(dev), K.conv_bias_to_engine(b, (1, 2, 2)).to(dev),
                   mode=2, stride=(1, 2, 2), pad_zero=True)
    assert out.shape == (T, 2 * H, 2 * W, C)
    assert rel_l2(out.float().cpu(), ref) < 6e-3


@pytest.mark.parametrize("P,C,G,with_res", [(90, 64, 32, False), (1000, 1024, 32, True), (37, 128, 8, True)])
def test_groupnorm_silu(K, dev, P, C, G, with_res):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(P + C)
    x = q(torch.randn(P, C, generator=g) * 2 + 0.5)
    res = q(torch.randn(P, C, generator=g)) if with_res else None
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    y = F.group_norm(x.t()[None], G, gamma, beta, 1e-5)[0].t()          # stats over (C/G, all positions)
    ref = F.silu(y + (res if with_res else 0))
    out = K.groupnorm_silu(x.to(dev, BF), gamma.to(dev), beta.to(dev), G, res=res.to(dev, BF) if with_res else None)
    assert rel_l2(out.float().cpu(), ref) < 6e-3


def test_conv3d_baseline_stage_size(K, dev):
    """The VAE decoder's heaviest stage geometry (H=128, W=192, 128 -> 128 channels at 768x512) on 5 frames against
    the oracle conv: exercises the multi-row-tile grid and reflect/replicate taps at full spatial size."""
    from oracle import vae
    T, H, W, C = 5, 128, 192, 128
    g = torch.Generator().manual_seed(11)
    x = q(torch.randn(1, C, T, H, W, generator=g))
    w = q(torch.randn(C, C, 3, 3, 3, generator=g) / math.sqrt(27 * C))
    b = torch.randn(C, generator=g)
    ref = vae.conv3d_simple(x, w, b, causal=False)[0].permute(1, 2, 3, 0)
    out = K.conv3d(x[0].permute(1, 2, 3, 0).contiguous().to(dev, BF), K.conv_weight_to_engine(w).to(dev), b.to(dev))
    assert rel_l2(out.float().cpu(), ref) < 6e-3


@pytest.mark.parametrize("grid,dim,heads,max_pos", [((9, 16, 24), 4096, 32, [20, 2048, 2048]), ((2, 3, 4), 256, 2, [20, 2048, 2048])])
def test_rope_tables_gpu(K, dev, grid, dim, heads, max_pos):
    """GPU SPLIT-RoPE table builder vs the oracle's precompute_freqs_cis restatement (arguments up to ~1.6e4 rad)
    and the 1-D audio / cross-modal variant."""
    from oracle import dit, dit_av, loop
    pos = loop.video_positions(1, *grid, 24.0)
    cos, sin = K.rope_tables(pos.to(dev), dim, 10000.0, max_pos)
    rc, rs = dit.rope_split_tables(pos, dim, heads, 10000.0, max_pos)          # [1, H, N, d/2]
    rc, rs = rc[0].permute(1, 0, 2).reshape(cos.shape), rs[0].permute(1, 0, 2).reshape(sin.shape)
    assert (cos.cpu() - rc).abs().max() < 2e-5 and (sin.cpu() - rs).abs().max() < 2e-5
    apos = dit_av.audio_positions(1, 37)
    c1, s1 = K.rope_tables(apos.to(dev), 2048, 10000.0, [20])
    r1c, r1s = dit.rope_split_tables(apos, 2048, 32, 10000.0, [20])
    assert (c1.cpu() - r1c[0].permute(1, 0, 2).reshape(c1.shape)).abs().max() < 2e-5
    assert (s1.cpu() - r1s[0].permute(1, 0, 2).reshape(s1.shape)).abs().max() < 2e-5


# ------------------------------------------------------------------------------------------ fp8 compute path (BASELINE config 3)
def quant_rows_emulated(x_bf16):
    """CPU emulation of ltx2_quantize_rows_fp8 with torch's IEEE fp32 arithmetic and its float8_e4m3fn cast (RNE, OCP)."""
    xf = x_bf16.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = 1.0 / scale
    return (xf * inv[:, None]).to(torch.float8_e4m3fn).view(torch.uint8), scale


def test_fp8_quantiser_is_bit_exact(dev):
    """The per-row e4m3fn quantiser against its fp32 / float8_e4m3fn emulation on the host, bit for bit: random rows, rows with
    outliers, tiny values (subnormal codes), an all-zero row, the DiT's K values."""
    g = torch.Generator().manual_seed(5)
    for K in (128, 4096, 16384):
        x = torch.randn(37, K, generator=g)
        x[3] *= 1e-3
        x[5, 7] = 300.0
        x[6] = 0.0
        x[7] *= 1e4
        x = x.to(torch.bfloat16)
        import ltx_2_mlx_amd.kernels as KK
        codes, scale = KK.quantize_rows_fp8(x.to(dev))
        rc, rs = quant_rows_emulated(x)
        assert torch.equal(scale.cpu(), rs), K
        assert torch.equal(codes.cpu(), rc), (K, int((codes.cpu() != rc).sum()))


@pytest.mark.parametrize("M,N,K", [(3456, 1024, 4096), (1100, 512, 512), (256, 256, 16384), (3456, 4096, 1024)])
def test_gemm_fp8_matches_dequantised_product(dev, M, N, K):
    """ltx2_gemm_fp8 = ascale[m] wscale[n] sum_k a8 w8 + bias on the fp8 MFMA against the same product of the DEQUANTISED codes in fp64
    (the codes are exact in fp32, so the only difference is fp32 accumulation order), asymmetric operands, ragged M; every epilogue."""
    import ltx_2_mlx_amd.kernels as KK
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * (1 + torch.arange(M)[:, None] % 7)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    a8, asc = KK.quantize_rows_fp8(a)
    w8, wsc = KK.quantize_rows_fp8(w)
    ref = (a8.view(torch.float8_e4m3fn).double() @ w8.view(torch.float8_e4m3fn).double().t()) * asc.double()[:, None] * wsc.double()[None, :] + bias.double()
    out = KK.gemm_fp8(a8, asc, w8, wsc, bias, epilogue=nv.EPI_F32)
    assert rel_l2(out.cpu(), ref.cpu()) < 5e-5
    # and the quantised product tracks the bf16 one to fp8 accuracy (3 mantissa bits on both sides)
    full = a.double() @ w.double().t() + bias.double()
    assert rel_l2(out.cpu(), full.cpu()) < 6e-2
    ob = KK.gemm_fp8(a8, asc, w8, wsc, bias, epilogue=nv.EPI_BF16)
    assert rel_l2(ob.cpu().float(), ref.cpu()) < 4e-3
    og = KK.gemm_fp8(a8, asc, w8, wsc, bias, epilogue=nv.EPI_GELU_BF16)
    assert rel_l2(og.cpu().float(), torch.nn.functional.gelu(ref.float(), approximate="tanh").cpu()) < 6e-3
    x = torch.randn(M, N, generator=g).to(dev)
    gt = torch.randn(N, generator=g).to(dev)
    x2 = x.clone()
    KK.gemm_fp8(a8, asc, w8, wsc, bias, epilogue=nv.EPI_RESID_GATE_F32, out=x2, gate_table=gt)
    assert rel_l2(x2.cpu(), (x.double() + gt.double()[None] * ref).cpu()) < 5e-5
    assert torch.equal(KK.gemm_fp8(a8, asc, w8, wsc, bias, epilogue=nv.EPI_F32), out)        # bit-reproducible


def test_norm_fp8_fusion_is_bit_identical(dev):
    """norm_mod with the fused per-token quantiser == norm_mod (bf16) followed by ltx2_quantize_rows_fp8, bit for bit: row-invariant
    modulation (the grid-stride kernel), per-token modulation and the plain norm (the row-per-block kernel), D = 4096 and 2048."""
    import ltx_2_mlx_amd.kernels as KK
    g = torch.Generator().manual_seed(8)
    for rows, D in ((3456, 4096), (1030, 2048)):
        x = (torch.randn(rows, D, generator=g) * (1 + 5 * torch.rand(rows, 1, generator=g))).to(dev)
        tab, emb = (0.1 * torch.randn(2, D, generator=g)).to(dev), (0.1 * torch.randn(2, D, generator=g)).to(dev)
        embt = (0.1 * torch.randn(rows, 2 * D, generator=g)).to(dev)
        cases = [dict(scale_tab=tab[1], shift_tab=tab[0], scale_emb=emb[1], shift_emb=emb[0]),                      # row-invariant
                 dict(scale_tab=tab[1], shift_tab=tab[0], scale_emb=embt[:, D:], shift_emb=embt[:, :D], emb_stride=2 * D),   # per token
                 dict()]                                                                                           # plain RMS norm
        for kw in cases:
            ref = KK.adaln_rmsnorm(x, **kw)
            rc, rs = KK.quantize_rows_fp8(ref)
            out, codes, scale = KK.adaln_rmsnorm_fp8(x, **kw)
            assert torch.equal(out, ref) and torch.equal(codes, rc) and torch.equal(scale, rs)
            _, codes2, scale2 = KK.adaln_rmsnorm_fp8(x, want_bf16=False, **kw)
            assert torch.equal(codes2, rc) and torch.equal(scale2, rs)


def test_gemm_fp8_fused_vt_matches_transpose_pass(dev):
    """The fp8 QKV projection writes attention's V^T operand from its epilogue like the bf16 kernel does: identical to the same GEMM
    followed by ltx2_vt_transpose."""
    import ltx_2_mlx_amd.kernels as KK
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator().manual_seed(9)
    M, H, hd = 3456, 8, 128
    D = H * hd
    a = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(torch.bfloat16).to(dev)
    b = torch.randn(3 * D, generator=g).to(dev)
    a8, asc = KK.quantize_rows_fp8(a)
    w8, wsc = KK.quantize_rows_fp8(w)
    full = KK.gemm_fp8(a8, asc, w8, wsc, b, epilogue=nv.EPI_BF16)
    vt_ref = KK.vt_transpose(full[:, 2 * D:], H, head_dim=hd)
    out, vt, fused = KK.gemm_fp8_qkv_vt(a8, asc, w8, wsc, b, H, hd)
    assert fused and torch.equal(out[:, :2 * D], full[:, :2 * D]) and torch.equal(vt, vt_ref)


def test_qnorm_folded_into_text_cross_attention(dev):
    """q_norm as arithmetic (VERDICT r2 #4c): the query projection's epilogue writes the partial sums of squares of its rows, the attention
    kernel turns them into a per-row softmax scale, q_norm.weight rides on the keys -- against q_norm applied to Q first (the pass this
    replaces), at the DiT's cross-attention geometry (3456 queries x 32 heads x 128, 1024 keys) and a ragged one."""
    import ltx_2_mlx_amd.kernels as KK
    g = torch.Generator().manual_seed(12)
    for Nq, S, H in ((3456, 1024, 32), (1100, 200, 8)):
        D = H * 128
        x = torch.randn(Nq, D, generator=g).to(torch.bfloat16).to(dev)
        wq = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(torch.bfloat16).to(dev)
        bq = (0.1 * torch.randn(D, generator=g)).to(dev)
        qn, kn = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
        k = torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev)
        v = torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev)
        vt = KK.vt_transpose(v, H)
        q, rowss = KK.gemm_rowss(x, wq, bq)
        assert torch.equal(q, KK.gemm(x, wq, bq))
        if D < 256 * 8:           # small grids stay on the 128x128 tile: no partial sums, the caller keeps the q_norm pass
            assert rowss is None
            continue
        ref_ss = (q.float() ** 2).reshape(Nq, D // 64, 64).sum(-1)
        assert rel_l2(rowss.cpu(), ref_ss.cpu()) < 1e-6
        # reference: q_norm applied to Q (the qknorm pass), k_norm on K; folded: raw q, K carries kn * qn, per-row scale from the partial sums
        qa, ka = q.clone(), k.clone()
        KK.qknorm_rope_(qa, D, 128, 0, qn)
        KK.qknorm_rope_(ka, D, 128, 0, kn)
        ref = KK.flash_attn(qa, ka, vt, H, S)
        kb = k.clone()
        KK.qknorm_rope_(kb, D, 128, 0, kn * qn)
        out = KK.flash_attn_rowscale(q, kb, vt, H, S, rowss)
        assert rel_l2(out.float().cpu(), ref.float().cpu()) < 6e-3
        # and against fp64 math
        qf = q.double()
        qf = qf * torch.rsqrt((qf * qf).mean(-1, keepdim=True) + 1e-6) * qn.double()
        kf = k.double()
        kf = (kf * torch.rsqrt((kf * kf).mean(-1, keepdim=True) + 1e-6) * kn.double()).to(torch.bfloat16).double()     # the cached keys are 16-bit
        qh, kh, vh = [t.reshape(-1, H, 128).transpose(0, 1) for t in (qf, kf, v.double())]
        exact = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128), dim=-1) @ vh).transpose(0, 1).reshape(Nq, D)
        assert rel_l2(out.double().cpu(), exact.cpu()) < 6e-3 and rel_l2(out.double().cpu(), exact.cpu()) <= rel_l2(ref.double().cpu(), exact.cpu()) * 1.2


@pytest.mark.parametrize("hd,H,Nq,S", [(128, 4, 300, 200), (64, 8, 130, 333), (128, 32, 512, 1024)])
def test_flash_attention_key_mask(dev, hd, H, Nq, S):
    """Boolean key mask of the text cross-attention (reference attention.py:38-70 fed by model.py:163-201): against fp64 softmax with the
    reference's additive -3.4e38 bias -- random holes, a padded tail, the whole FIRST KV tile masked, one key left, and every key masked
    (the reference's bias then cancels in the softmax: the mean of V over ALL keys)."""
    import ltx_2_mlx_amd.kernels as KK
    g = torch.Generator().manual_seed(77)
    D = H * hd
    q = torch.randn(Nq, D, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev)
    vt = KK.vt_transpose(v, H, head_dim=hd)
    masks = {"holes": torch.rand(S, generator=g) > 0.4, "tail": torch.arange(S) < S // 3, "first_tile": torch.arange(S) >= 64,
             "one_key": torch.arange(S) == S - 1, "none": torch.zeros(S, dtype=torch.bool), "all": torch.ones(S, dtype=torch.bool)}
    qh, kh, vh = [t.double().cpu().reshape(-1, H, hd).transpose(0, 1) for t in (q, k, v)]
    for tag, mk in masks.items():
        bias = (1 - mk.double()) * -3.40e38
        s = (qh @ kh.transpose(1, 2)).float() / math.sqrt(hd) + bias.float()           # fp32 like the reference: the bias absorbs the score
        ref = (torch.softmax(s.double(), dim=-1) @ vh).transpose(0, 1).reshape(Nq, D)
        out = KK.flash_attn_keymask(q, k, vt, H, S, mk.to(torch.int32) if tag == "holes" else mk)
        assert torch.isfinite(out).all(), tag
        assert rel_l2(out.double().cpu(), ref) < 6e-3, (tag, rel_l2(out.double().cpu(), ref))
    # an all-ones mask against the unmasked kernel: the masked form re-references every tile, the unmasked one keeps its first tile's maximum -- two
    # roundings of P to 16 bits around different exponents (each ~2.5e-3 from the exact result)
    assert rel_l2(KK.flash_attn_keymask(q, k, vt, H, S, masks["all"]).float().cpu(), KK.flash_attn(q, k, vt, H, S).float().cpu()) < 6e-3


# ------------------------------------------------------------------------------------------ round 4: the AudioVideo block's cross-modal section
@pytest.mark.parametrize("rows,D", [(3456, 4096), (68, 2048), (5, 512), (130, 6144)])
def test_adaln_rmsnorm2_equals_two_passes(K, dev, rows, D):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g).to(dev)
    t = [(0.1 * torch.randn(D, generator=g)).to(dev) for _ in range(4)]
    o0, o1 = K.adaln_rmsnorm2(x, t[0], t[1], t[2], t[3])
    r0 = K.adaln_rmsnorm(x, 1e-6, False, t[0], t[1])
    r1 = K.adaln_rmsnorm(x, 1e-6, False, t[2], t[3])
    assert torch.equal(o0, r0) and torch.equal(o1, r1)
    p0, p1 = K.adaln_rmsnorm2(x, None, None, t[2], None)
    assert torch.equal(p0, K.adaln_rmsnorm(x)) and torch.equal(p1, K.adaln_rmsnorm(x, 1e-6, False, t[2], None))


@pytest.mark.parametrize("heads,hd,Nq,Nkv", [(32, 128, 1100, 1024), (32, 64, 68, 3456), (4, 64, 300, 68)])
def test_flash_attn_gated_epilogue(K, dev, heads, hd, Nq, Nkv):
    """Per-head gates folded into the attention epilogue (round 4) against attention followed by the gating pass it replaces, and fp64."""
    g = torch.Generator().manual_seed(Nq + Nkv)
    D = heads * hd
    qq = torch.randn(Nq, D, generator=g).to(BF).to(dev)
    kk = torch.randn(Nkv, D, generator=g).to(BF).to(dev)
    vv = torch.randn(Nkv, D, generator=g).to(BF).to(dev)
    logits = (2.0 * torch.randn(Nq, heads, generator=g)).to(dev)
    vt = K.vt_transpose(vv, heads, head_dim=hd)
    plain = K.flash_attn(qq, kk, vt, heads, Nkv)
    out = K.flash_attn_gated(qq, kk, vt, heads, Nkv, logits)
    gate = (2 * torch.sigmoid(logits.double()))[..., None]
    two_pass = (plain.double().reshape(Nq, heads, hd) * gate).reshape(Nq, D)
    assert rel_l2(out.double().cpu(), two_pass.cpu()) < 4e-3
    qh, kh, vh = [t.double().reshape(-1, heads, hd).transpose(0, 1) for t in (qq, kk, vv)]
    exact = ((torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh).transpose(0, 1) * gate).reshape(Nq, D)
    assert rel_l2(out.double().cpu(), exact.cpu()) < 6e-3



@pytest.mark.parametrize("heads,hd,Nq,Nkv,Dq", [(32, 128, 3456, 1024, 4096), (8, 128, 1111, 300, 1024), (16, 64, 1500, 68, 2048), (32, 64, 100, 200, 2048)])
def test_flash_attn_gated_from_k_slice_partial_logits(K, dev, heads, hd, Nq, Nkv, Dq):
    """the engine's many-row gate path (gate_logits_parts_kernel: 8 K-slice partial sums of x @ Wg^T, summed with the bias in the attention epilogue) had
    no kernel-level test.  Against ltx2_attn_head_gate's one-launch logits + the gated attention, and fp64: ragged M, H < 16, H = 32, few rows."""
    g = torch.Generator().manual_seed(heads * 1000 + Nq)
    D = heads * hd
    qq = torch.randn(Nq, D, generator=g).to(BF).to(dev)
    kk = torch.randn(Nkv, D, generator=g).to(BF).to(dev)
    vv = torch.randn(Nkv, D, generator=g).to(BF).to(dev)
    x = torch.randn(Nq, Dq, generator=g).to(BF).to(dev)
    wg = (torch.randn(heads, Dq, generator=g) / math.sqrt(Dq)).to(BF).to(dev)
    bg = (0.5 * torch.randn(heads, generator=g)).to(dev)
    vt = K.vt_transpose(vv, heads, head_dim=hd)
    out = K.flash_attn_gated_parts(qq, kk, vt, heads, Nkv, x, wg, bg)
    scratch = torch.ones(Nq, D, device=dev, dtype=BF)
    logits = K.attn_head_gate_(scratch, x, wg, bg, heads)                                   # the one-launch kernel's logits
    ref = K.flash_attn_gated(qq, kk, vt, heads, Nkv, logits)
    assert rel_l2(out.float().cpu(), ref.float().cpu()) < 7e-5                               # same gates up to the summation order of the logits (measured 1.0e-5 ... 1.4e-5)
    lg = x.double() @ wg.double().T + bg.double()
    gate = (2 * torch.sigmoid(lg))[..., None]
    qh, kh, vh = [t.double().reshape(-1, heads, hd).transpose(0, 1) for t in (qq, kk, vv)]
    exact = ((torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(hd), dim=-1) @ vh).transpose(0, 1) * gate).reshape(Nq, D)
    assert rel_l2(out.double().cpu(), exact.cpu()) < 6e-3
    assert torch.equal(out, K.flash_attn_gated_parts(qq, kk, vt, heads, Nkv, x, wg, bg))


@pytest.mark.parametrize("M,D,NO,must", [(3456, 4096, 4096, True), (13824, 4096, 4096, True), (1300, 1024, 4096, False), (600, 512, 512, False)])
def test_norm_folded_around_the_gemms(dev, M, D, NO, must):
    """Round 6 (GemmParams::shadow / rf_parts): rms_norm(x) (1 + s) in front of a projection (W, b) as r (x (1 + s)) W^T + b.  Producer = a gated-residual
    GEMM that also leaves y = bf16(x_new (1 + s)) and the partial sums of squares of x_new per 256-column tile; consumer = the projection that forms the row
    factors from those partials inside the kernel.  Checked against the unfolded kernels (norm pass + plain GEMMs) and fp64."""
    import ltx_2_mlx_amd.kernels as KK
    from ltx_2_mlx_amd import _native as nv
    g = torch.Generator().manual_seed(606)
    eps = 1e-6
    att = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    wo = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(torch.bfloat16).to(dev)
    bo = (0.1 * torch.randn(D, generator=g)).to(dev)
    gate = (1.0 + 0.2 * torch.randn(D, generator=g)).to(dev)
    x0 = (3.0 * torch.randn(M, D, generator=g)).to(dev)
    sc = (0.3 * torch.randn(D, generator=g)).to(dev)
    wp = (torch.randn(NO, D, generator=g) / math.sqrt(D)).to(torch.bfloat16).to(dev)
    bp = (0.1 * torch.randn(NO, generator=g)).to(dev)
    for scale in (None, sc):
        # ---- unfolded: x += gate (att Wo^T + bo); h = norm(x)(1 + scale); out = h Wp^T + bp
        xa = x0.clone()
        KK.gemm(att, wo, bo, nv.EPI_RESID_GATE_F32, out=xa, gate_table=gate)
        h = KK.adaln_rmsnorm(xa, eps, scale_tab=scale)
        ref = KK.gemm(h, wp, bp)
        # ---- folded
        xb = x0.clone()
        y = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        r = KK.gemm_fold(att, wo, bo, nv.EPI_RESID_GATE_F32, out=xb, gate_table=gate, shadow=y, shadow_scale=scale)
        if r is None:           # the dispatch keeps this shape off the 4-wave kernel: the engine then runs the norm pass (fold_supported())
            assert not must
            return
        _, ss = r
        assert torch.equal(xa, xb)                                              # the residual stream itself is untouched by the extra outputs
        assert rel_l2(ss[:, :M].sum(0).cpu(), (xa.double() ** 2).sum(-1).cpu()) < 1e-6
        s1 = 1.0 if scale is None else (1.0 + scale)
        assert torch.equal(y, (xa * s1).to(torch.bfloat16))                     # the shadow: one rounding of x (1 + s)
        out = KK.gemm_fold(y, wp, bp, nv.EPI_BF16, rf_parts=ss, rf_dim=D, eps=eps)
        exact = ((xa.double() * torch.rsqrt((xa.double() ** 2).mean(-1, keepdim=True) + eps)) * (1 if scale is None else (1 + scale.double()))) @ wp.double().T + bp.double()
        e_fold, e_ref = rel_l2(out.double().cpu(), exact.cpu()), rel_l2(ref.double().cpu(), exact.cpu())
        assert e_fold < 6e-3 and e_fold < 1.5 * e_ref + 1e-4                     # the same accuracy class as norm-then-GEMM
        # the GELU consumer with the same row factors: against gelu of the fp64 product
        og = KK.gemm_fold(y, wp, bp, nv.EPI_GELU_BF16, rf_parts=ss, rf_dim=D, eps=eps)
        eg = torch.nn.functional.gelu(exact.float(), approximate="tanh")
        assert rel_l2(og.float().cpu(), eg.cpu()) < 8e-3
