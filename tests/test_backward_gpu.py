"""Op-level parity of the backward building blocks (uni_renderer_amd/backward.py, csrc/backward.hip; SURVEY section 8a,
device op 11) against PyTorch autograd in fp32 on the CPU, on the same rounded inputs.  Tolerances: rel-L2 of the
fp16 / bf16 result against the fp32 gradient."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 3e-3, torch.bfloat16: 2e-2}


def _rand(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_and_colsum(dev, dtype):
    from uni_renderer_amd import backward as bw
    x = _rand((3, 200, 136), dtype, dev, 1)
    t = bw.transpose2d(x)
    assert t.shape == (3, 136, 200) and torch.equal(t, x.transpose(1, 2).contiguous())
    s = bw.colsum(x)
    assert rel_l2(s, x.float().cpu().reshape(-1, 136).sum(0)) < 1e-5
    sg = bw.colsum(x, rows_per_group=200)
    assert rel_l2(sg, x.float().cpu().sum(1)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_backward(dev, dtype):
    from uni_renderer_amd import backward as bw
    x = _rand((2, 100, 128), dtype, dev, 1)
    w = _rand((192, 128), dtype, dev, 2, 0.1)
    dy = _rand((2, 100, 192), dtype, dev, 3)
    dx, dw, db = bw.linear_backward(x, w, dy)
    xr, wr = x.float().cpu().requires_grad_(), w.float().cpu().requires_grad_()
    br = torch.zeros(192, requires_grad=True)
    (F.linear(xr, wr, br) * dy.float().cpu()).sum().backward()
    assert rel_l2(dx, xr.grad) < TOL[dtype]
    assert rel_l2(dw, wr.grad) < TOL[dtype]
    assert rel_l2(db, br.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_backward(dev, dtype):
    from uni_renderer_amd import backward as bw
    from uni_renderer_amd.layers import pack_conv3x3
    B, H, W, Cin, Cout = 2, 10, 12, 64, 128
    x = _rand((B, H, W, Cin), dtype, dev, 1)
    w_nchw = _rand((Cout, Cin, 3, 3), dtype, dev, 2, 0.05)
    dy = _rand((B, H, W, Cout), dtype, dev, 3)
    dx, dw, db = bw.conv3x3_backward(x, pack_conv3x3(w_nchw, dtype), dy)
    xr = x.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
    wr = w_nchw.float().cpu().requires_grad_()
    br = torch.zeros(Cout, requires_grad=True)
    (F.conv2d(xr, wr, br, padding=1) * dy.float().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert rel_l2(dx, xr.grad.permute(0, 2, 3, 1)) < TOL[dtype]
    # packed layout [N][(ky, kx, c)]
    assert rel_l2(dw, wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)) < TOL[dtype]
    assert rel_l2(db, br.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_and_geglu_backward(dev, dtype):
    from uni_renderer_amd import backward as bw
    x = _rand((4, 50, 64), dtype, dev, 1, 2.0)
    dy = _rand((4, 50, 64), dtype, dev, 2)
    xr = x.float().cpu().requires_grad_()
    (F.silu(xr) * dy.float().cpu()).sum().backward()
    assert rel_l2(bw.silu_backward(x, dy), xr.grad) < TOL[dtype]
    h = _rand((3, 40, 256), dtype, dev, 3, 1.5)
    dyo = _rand((3, 40, 128), dtype, dev, 4)
    hr = h.float().cpu().requires_grad_()
    a, g = hr.chunk(2, dim=-1)
    y = a * F.gelu(g)
    assert rel_l2(bw.geglu_forward(h), y) < TOL[dtype]
    (y * dyo.float().cpu()).sum().backward()
    assert rel_l2(bw.geglu_backward(h, dyo), hr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("C", [64, 320, 1280])
def test_groupnorm_backward(dev, dtype, silu, C):
    from uni_renderer_amd import backward as bw
    x = _rand((2, 9, 7, C), dtype, dev, 1, 1.5) + 0.3
    dy = _rand((2, 9, 7, C), dtype, dev, 2)
    gam = torch.randn(C, generator=torch.Generator().manual_seed(3)).to(dev)
    bet = torch.randn(C, generator=torch.Generator().manual_seed(4)).to(dev)
    dx, dg, db = bw.groupnorm_backward(x, dy, gam, bet, 1e-5, groups=32, silu=silu)
    xr = x.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
    gr, br = gam.cpu().clone().requires_grad_(), bet.cpu().clone().requires_grad_()
    y = F.group_norm(xr, 32, gr, br, 1e-5)
    if silu:
        y = F.silu(y)
    (y * dy.float().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert rel_l2(dx, xr.grad.permute(0, 2, 3, 1)) < TOL[dtype] * 2
    assert rel_l2(dg, gr.grad) < TOL[dtype]
    assert rel_l2(db, br.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("shape", [(4, 32, 32, 640), (1, 16, 16, 960), (2, 8, 8, 2560), (3, 16, 8, 1920), (2, 32, 32, 320)])
def test_groupnorm_backward_one_launch_against_the_chunked_path(dev, dtype, silu, shape, monkeypatch):
    """ur_groupnorm_backward_fused (one workgroup per (sample, group): the 32x32 / 16x16 / 8x8 maps of the training step, group
    widths 10 .. 80 = pieces of 2 / 4 / 8 channels) against fp32 autograd and against the four-launch path on the same inputs."""
    from uni_renderer_amd import backward as bw
    B, H, W, C = shape
    x = _rand((B, H, W, C), dtype, dev, 1, 1.5) + 0.3
    dy = _rand((B, H, W, C), dtype, dev, 2)
    gam = torch.randn(C, generator=torch.Generator().manual_seed(3)).to(dev)
    bet = torch.randn(C, generator=torch.Generator().manual_seed(4)).to(dev)
    assert H * W <= bw.GN_BWD_FUSED_MAX_ROWS
    dx, dg, db = bw.groupnorm_backward(x, dy, gam, bet, 1e-5, groups=32, silu=silu)
    monkeypatch.setattr(bw, "GN_BWD_FUSED_MAX_ROWS", 0)
    dx2, dg2, db2 = bw.groupnorm_backward(x, dy, gam, bet, 1e-5, groups=32, silu=silu)
    xr = x.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()
    gr, br = gam.cpu().clone().requires_grad_(), bet.cpu().clone().requires_grad_()
    y = F.group_norm(xr, 32, gr, br, 1e-5)
    if silu:
        y = F.silu(y)
    (y * dy.float().cpu().permute(0, 3, 1, 2)).sum().backward()
    for got, ref, tol in ((dx, xr.grad.permute(0, 2, 3, 1), 2.0), (dg, gr.grad, 1.0), (db, br.grad, 1.0)):
        assert rel_l2(got, ref) < TOL[dtype] * tol
    assert rel_l2(dx, dx2.float().cpu()) < TOL[dtype] and rel_l2(dg, dg2.cpu()) < 1e-4 and rel_l2(db, db2.cpu()) < 1e-4
    dx3, dg3, db3 = (lambda: (monkeypatch.setattr(bw, "GN_BWD_FUSED_MAX_ROWS", 1024), bw.groupnorm_backward(x, dy, gam, bet, 1e-5, groups=32, silu=silu))[1])()
    assert torch.equal(dx, dx3) and torch.equal(dg, dg3) and torch.equal(db, db3)   # fixed-order sums


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm_backward(dev, dtype, C):
    from uni_renderer_amd import backward as bw
    x = _rand((3, 111, C), dtype, dev, 1, 2.0) + 0.5
    dy = _rand((3, 111, C), dtype, dev, 2)
    gam = torch.randn(C, generator=torch.Generator().manual_seed(3)).to(dev)
    dx, dg, db = bw.layernorm_backward(x, dy, gam, 1e-5)
    xr = x.float().cpu().requires_grad_()
    gr = gam.cpu().clone().requires_grad_()
    br = torch.zeros(C, requires_grad=True)
    (F.layer_norm(xr, (C,), gr, br, 1e-5) * dy.float().cpu()).sum().backward()
    assert rel_l2(dx, xr.grad) < TOL[dtype] * 2
    assert rel_l2(dg, gr.grad) < TOL[dtype]
    assert rel_l2(db, br.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2, 40, 128, 128), (1, 8, 40, 200, 77), (2, 4, 80, 64, 64), (1, 2, 160, 96, 96)])
def test_attention_backward(dev, dtype, shape):
    from uni_renderer_amd import backward as bw
    B, H, d, Tq, Tk = shape
    C = H * d
    q = _rand((B, Tq, C), dtype, dev, 1)
    k = _rand((B, Tk, C), dtype, dev, 2)
    v = _rand((B, Tk, C), dtype, dev, 3)
    do = _rand((B, Tq, C), dtype, dev, 4)
    dq, dk, dv = bw.attention_backward(q, k, v, do, H)
    qr, kr, vr = (t.float().cpu().requires_grad_() for t in (q, k, v))
    o = F.scaled_dot_product_attention(qr.view(B, Tq, H, d).transpose(1, 2), kr.view(B, Tk, H, d).transpose(1, 2),
                                       vr.view(B, Tk, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, Tq, C)
    (o * do.float().cpu()).sum().backward()
    tol = TOL[dtype] * 2  # P and dS are materialised in the compute dtype
    assert rel_l2(dq, qr.grad) < tol and rel_l2(dk, kr.grad) < tol and rel_l2(dv, vr.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2, 40, 128), (1, 3, 40, 192), (2, 2, 80, 64), (1, 2, 160, 128), (1, 2, 32, 64),
                                   (1, 8, 40, 1024)])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("direct", [None, 0, 1000])
def test_flash_attention_backward(dev, dtype, shape, fused, direct, monkeypatch):
    """ur_attention_backward (P in registers, two launches) against fp32 autograd through SDPA, with the forward output
    of the flash kernel as ``o``; head dims of all three levels (40 / 80 / 160 -> padded 64 / 96 / 160), both query
    widths per wave (T % 128 == 0 or not), separate q / k / v and the fused [B, T, 3C] layout.  Also against the
    materialised-P path (the two must agree to rounding of P / dS)."""
    from uni_renderer_amd import backward as bw
    from uni_renderer_amd import autograd_ops as A
    if direct is not None:  # 0: every head dim reads the [B, T, H*d] layout in place; 1000: every head dim through copies
        monkeypatch.setattr(bw, "FLASH_DIRECT_MIN_D", direct)
    B, H, d, T = shape
    C = H * d
    q, k, v = (_rand((B, T, C), dtype, dev, i + 1) for i in range(3))
    do = _rand((B, T, C), dtype, dev, 4)
    qr, kr, vr = (t.float().cpu().requires_grad_() for t in (q, k, v))
    o_ref = F.scaled_dot_product_attention(qr.view(B, T, H, d).transpose(1, 2), kr.view(B, T, H, d).transpose(1, 2),
                                           vr.view(B, T, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, T, C)
    (o_ref * do.float().cpu()).sum().backward()
    assert bw.FLASH_BACKWARD and bw._lib.load().ur_attention_backward_supported(T, T, d)
    if fused:
        qkv = torch.cat([q, k, v], dim=-1).requires_grad_()
        o = A.AttentionQKV.apply(qkv, H)
        o.backward(do)
        dq, dk, dv = qkv.grad.split(C, dim=-1)
        m = bw.attention_backward(qkv.detach(), qkv.detach(), qkv.detach(), do, H, fused_qkv=True).split(C, dim=-1)
    else:
        q_, k_, v_ = (t.clone().requires_grad_() for t in (q, k, v))
        o = A.Attention.apply(q_, k_, v_, H)
        o.backward(do)
        dq, dk, dv = q_.grad, k_.grad, v_.grad
        m = bw.attention_backward(q, k, v, do, H)  # no ``o``: the materialised-P path
    assert rel_l2(o, o_ref) < TOL[dtype]
    tol = TOL[dtype] * 2  # P and dS enter the MFMA in the compute dtype
    errs = [rel_l2(dq, qr.grad), rel_l2(dk, kr.grad), rel_l2(dv, vr.grad)]
    print({"flash_attention_backward": str(dtype), "shape": shape, "fused": fused, "rel_l2_dq_dk_dv": errs,
           "vs_materialised": [rel_l2(a, b) for a, b in zip((dq, dk, dv), m)]})
    assert max(errs) < tol, errs
    for a, b in zip((dq, dk, dv), m):
        assert rel_l2(a, b) < tol


def test_flash_attention_backward_deterministic(dev):
    from uni_renderer_amd import backward as bw
    B, H, d, T = 1, 4, 40, 256
    q, k, v, do, o = (_rand((B, T, H * d), torch.bfloat16, dev, i + 1) for i in range(5))
    a = bw.attention_backward(q, k, v, do, H, o=o)
    b = bw.attention_backward(q, k, v, do, H, o=o)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", DTYPES)
def test_autograd_functions_conv_variants(dev, dtype):
    """Conv3x3 Function: stride 2 (zero-insertion dgrad), nearest-2x + conv, fused bias / time-embedding row add /
    residual, and a 28-channel output (conv_out), against nn.functional autograd."""
    from uni_renderer_amd import autograd_ops as A
    B, H, W, Ci, Co = 2, 8, 8, 64, 128
    x = _rand((B, H, W, Ci), dtype, dev, 1).requires_grad_()
    w = (_rand((Co, Ci, 3, 3), torch.float32, dev, 2, 0.05)).requires_grad_()
    b = _rand((Co,), torch.float32, dev, 3).requires_grad_()
    row = _rand((B, Co), dtype, dev, 4).requires_grad_()
    res = _rand((B, H, W, Co), dtype, dev, 5).requires_grad_()
    dy = _rand((B, H, W, Co), dtype, dev, 6)
    y = A.conv3x3(x, A.pack_conv_weight(w, dtype), b, res=res, rowadd=row)
    (y.float() * dy.float()).sum().backward()
    xr = x.detach().float().cpu().permute(0, 3, 1, 2).requires_grad_()
    wr = w.detach().to(dtype).float().cpu().requires_grad_()
    br = b.detach().cpu().requires_grad_()
    rr, sr = row.detach().float().cpu().requires_grad_(), res.detach().float().cpu().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1) + rr[:, :, None, None] + sr.permute(0, 3, 1, 2)
    (yr * dy.float().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert rel_l2(y, yr.permute(0, 2, 3, 1)) < TOL[dtype]
    assert rel_l2(x.grad, xr.grad.permute(0, 2, 3, 1)) < TOL[dtype]
    assert rel_l2(w.grad, wr.grad) < TOL[dtype]
    assert rel_l2(b.grad, br.grad) < 1e-3 and rel_l2(row.grad, rr.grad) < TOL[dtype] and rel_l2(res.grad, sr.grad) < 1e-6
    # stride 2 and upsample + conv, 28 output channels
    for mode in ("s2", "up"):
        x2 = _rand((B, H, W, Ci), dtype, dev, 7).requires_grad_()
        w2 = (_rand((28, Ci, 3, 3), torch.float32, dev, 8, 0.05)).requires_grad_()
        if mode == "s2":
            y2 = A.conv3x3(x2, A.pack_conv_weight(w2, dtype), None, stride=2)
        else:
            y2 = A.conv3x3(A.Up2x.apply(x2), A.pack_conv_weight(w2, dtype), None)
        g2 = _rand(tuple(y2.shape), dtype, dev, 9)
        (y2.float() * g2.float()).sum().backward()
        x2r = x2.detach().float().cpu().permute(0, 3, 1, 2).requires_grad_()
        w2r = w2.detach().to(dtype).float().cpu().requires_grad_()
        if mode == "s2":
            y2r = F.conv2d(x2r, w2r, None, stride=2, padding=1)
        else:
            y2r = F.conv2d(F.interpolate(x2r, scale_factor=2.0, mode="nearest"), w2r, None, padding=1)
        (y2r * g2.float().cpu().permute(0, 3, 1, 2)).sum().backward()
        assert rel_l2(y2, y2r.permute(0, 2, 3, 1)) < TOL[dtype], mode
        assert rel_l2(x2.grad, x2r.grad.permute(0, 2, 3, 1)) < TOL[dtype], mode
        assert rel_l2(w2.grad, w2r.grad) < TOL[dtype], mode


@pytest.mark.parametrize("dtype", DTYPES)
def test_autograd_functions_chain(dev, dtype):
    """LayerNorm -> q/k/v Linear -> Attention -> Linear(+res) -> GroupNorm(+SiLU) chained through autograd."""
    from uni_renderer_amd import autograd_ops as A
    B, T, C, H = 2, 64, 128, 2
    x = _rand((B, T, C), dtype, dev, 1).requires_grad_()
    par = {n: (_rand(s, torch.float32, dev, i + 10, sc)).requires_grad_() for i, (n, s, sc) in enumerate(
        [("lg", (C,), 1.0), ("lb", (C,), 1.0), ("wq", (C, C), 0.1), ("wk", (C, C), 0.1), ("wv", (C, C), 0.1), ("wo", (C, C), 0.1),
         ("bo", (C,), 1.0), ("gg", (C,), 1.0), ("gb", (C,), 1.0)])}
    dy = _rand((B, 8, 8, C), dtype, dev, 30)

    def run(x, p, hip):
        if hip:
            xn = A.LayerNorm.apply(x, p["lg"], p["lb"], 1e-5)
            q, k, v = (A.linear(xn, p[n].to(dtype)) for n in ("wq", "wk", "wv"))
            o = A.Attention.apply(q, k, v, H)
            y = A.linear(o, p["wo"].to(dtype), p["bo"], res=x)
            return A.GroupNorm.apply(y.view(B, 8, 8, C), p["gg"], p["gb"], 1e-5, 32, True)
        xn = F.layer_norm(x, (C,), p["lg"], p["lb"], 1e-5)
        q, k, v = (F.linear(xn, p[n]) for n in ("wq", "wk", "wv"))
        sp = lambda t: t.view(B, T, H, C // H).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, T, C)
        y = F.linear(o, p["wo"], p["bo"]) + x
        return F.silu(F.group_norm(y.view(B, 8, 8, C).permute(0, 3, 1, 2), 32, p["gg"], p["gb"], 1e-5)).permute(0, 2, 3, 1)

    out = run(x, par, True)
    (out.float() * dy.float()).sum().backward()
    xr = x.detach().float().cpu().requires_grad_()
    pr = {n: (t.detach().to(dtype).float().cpu() if t.dim() == 2 else t.detach().cpu()).requires_grad_() for n, t in par.items()}
    outr = run(xr, pr, False)
    (outr * dy.float().cpu()).sum().backward()
    assert rel_l2(out, outr) < TOL[dtype] * 2
    assert rel_l2(x.grad, xr.grad) < TOL[dtype] * 4
    for n in par:
        assert rel_l2(par[n].grad, pr[n].grad) < TOL[dtype] * 4, n


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(128, 64, None), (64, 28, 64), (320, 4, 64), (64, 640, None)])
def test_conv_weight_pack_unpack_and_rotation_kernels(dev, dtype, shape):
    """ur_pack_conv_weight / ur_unpack_conv_weight_grad (fp32 master <-> packed compute-dtype matrix, one pass each way)
    against layers.pack_conv3x3 and its autograd; backward._rot_weights as ONE ur_transpose2d launch against the
    flip / permute expression it replaces."""
    from uni_renderer_amd import autograd_ops as A
    from uni_renderer_amd import backward as bw
    from uni_renderer_amd.layers import pack_conv3x3

    co, ci, cpad = shape
    g = torch.Generator().manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3, generator=g).to(dev).requires_grad_(True)
    wp = A.pack_conv_weight(w, dtype, cpad)
    ref = pack_conv3x3(w.detach(), dtype, cpad)
    assert wp.shape == ref.shape and torch.equal(wp, ref)
    up = torch.randn(wp.shape, generator=g).to(dev).to(dtype)
    wp.backward(up)
    cp = ci if cpad is None else cpad
    want = up.float().view(co, 3, 3, cp)[..., :ci].permute(0, 3, 1, 2)
    assert w.grad.shape == w.shape and w.grad.dtype == torch.float32 and torch.equal(w.grad, want.contiguous())
    if cp % 8 == 0 and co % 8 == 0:
        rot = bw._rot_weights(ref, cp)
        w4 = ref.view(co, 3, 3, cp)
        assert torch.equal(rot, w4.flip(1, 2).permute(3, 1, 2, 0).reshape(cp, 9 * co).contiguous())

def test_rotated_conv_weights_and_linear_transposes_in_multi_tensor_launches(dev):
    """backward.rot_weights_many (all dgrad conv weights of a network, 32 per ur_transpose2d_multi launch) against
    _rot_weights, and 40 matrices through transpose2d_many (two launches) against the single-tensor kernel."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 64), (128, 320), (320, 64), (64, 8)] * 9
    ws = [(torch.randn(co, 9 * ci, generator=g).to(torch.bfloat16).to(dev), ci) for co, ci in shapes]
    for (w, ci), r in zip(ws, bw.rot_weights_many(ws)):
        assert torch.equal(r, bw._rot_weights(w, ci))
    ms = [torch.randn(8 * (1 + i % 7), 16 * (1 + i % 5), generator=g).to(torch.bfloat16).to(dev) for i in range(40)]
    for m, t in zip(ms, bw.transpose2d_many(ms)):
        assert torch.equal(t, m.t().contiguous())


def test_split_and_merge_heads_of_several_tensors_in_one_launch(dev):
    """ur_split_heads_multi / ur_merge_heads_multi against the single-tensor entry points: the five per-head copies and the
    three gradients of a d = 40 attention backward, column offsets of a fused q | k | v, key rows padded to 128."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(21)
    B, H, d, dp, Tq, Tk, Tkp = 2, 8, 40, 64, 192, 77, 128
    qkv = torch.randn(B, Tq, 3 * H * d, generator=g).to(torch.bfloat16).to(dev)
    kv = torch.randn(B, Tk, 2 * H * d, generator=g).to(torch.bfloat16).to(dev)
    o = torch.randn(B, Tq, H * d, generator=g).to(torch.bfloat16).to(dev)
    items = [(qkv, Tq, 0), (kv, Tkp, 0), (kv, Tkp, H * d), (o, Tq, 0), (qkv, Tq, 2 * H * d)]
    many = bw.split_heads_many(items, H, d, dp)
    for (x, Tp, off), got in zip(items, many):
        assert torch.equal(got, bw._split_heads(x, H, d, Tp, dp, off))
    outs = [torch.zeros(B, Tq, 3 * H * d, dtype=torch.bfloat16, device=dev), torch.zeros(B, Tk, H * d, dtype=torch.bfloat16, device=dev)]
    bw.merge_heads_many([(many[0], outs[0], 0), (many[4], outs[0], 2 * H * d), (many[1], outs[1], 0)], B, H, d)
    assert torch.equal(outs[0][..., :H * d], qkv[..., :H * d]) and torch.equal(outs[0][..., 2 * H * d:], qkv[..., 2 * H * d:])
    assert torch.equal(outs[1], kv[..., :H * d]) and not outs[0][..., H * d:2 * H * d].any()


def test_column_sums_of_many_matrices_in_one_launch(dev):
    """backward.NormSums / ur_colsum_multi: the deferred gamma / beta gradients of the norm layers -- plain column sums
    (LayerNorm: per-wave rows [waves, 2 C]) and the (channel, component) pair form (GroupNorm: [B, C, 2] -> [2, C]), 100 items
    = two launches, against torch sums; two flushes give identical bits."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(33)
    q = bw.NormSums()
    items = []
    for i in range(100):
        M, N = (1 + 37 * i) % 2050 + 1, 2 * (8 + (13 * i) % 700)
        part = torch.randn(M, N, generator=g).to(dev)
        items.append((part, bool(i % 2), q.add(part, bool(i % 2))))
    q.flush()
    assert not q.items
    q2 = bw.NormSums()
    again = [q2.add(part, pair) for part, pair, _ in items]
    q2.flush()
    for (part, pair, out), out2 in zip(items, again):
        ref = part.double().sum(0)
        if pair:
            ref = ref.view(-1, 2).t().reshape(-1)
        assert (out.double() - ref).abs().max().item() <= 1e-4 * max(1.0, part.abs().sum(0).max().item())
        assert torch.equal(out, out2)


def test_transpose2d_many(dev):
    """ur_transpose2d_multi: several (batched, strided, ragged) transposes in one launch == the single-tensor kernel."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(5)
    big = torch.randn(3, 100, 3 * 64, generator=g).to(torch.bfloat16).to(dev)
    xs = [torch.randn(77, 320, generator=g).to(torch.bfloat16).to(dev), torch.randn(4, 130, 64, generator=g).to(torch.bfloat16).to(dev),
          big[..., 64:128], torch.randn(5, 8, generator=g).to(torch.bfloat16).to(dev),
          torch.randn(2, 3, 200, 72, generator=g).to(torch.bfloat16).to(dev)]
    outs = bw.transpose2d_many(xs)
    for x, o in zip(xs, outs):
        R = x.shape[-2]
        assert o.shape[-1] == (R + 7) // 8 * 8
        assert torch.equal(o[..., :R], x.transpose(-1, -2)) and float(o[..., R:].abs().sum()) == 0.0
        assert torch.equal(o, bw.transpose2d(x))

def test_transpose2d_many_pad64(dev):
    """rows_out: the launch itself zero-pads the transposed row length to the dW GEMM's 64-granularity"""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(308, 320, generator=g).to(torch.bfloat16).to(dev), torch.randn(4, 1280, generator=g).to(torch.bfloat16).to(dev),
          torch.randn(128, 64, generator=g).to(torch.bfloat16).to(dev)]
    outs = bw.transpose2d_many(xs, pad64=(0, 1))
    assert outs[0].shape == (320, 320) and outs[1].shape == (1280, 64) and outs[2].shape == (64, 128)
    for x, o in zip(xs, outs):
        R = x.shape[0]
        assert torch.equal(o[:, :R], x.t()) and float(o[:, R:].abs().sum()) == 0.0
    (o2, _, _), s = bw.transpose2d_many(xs, colsum_of=1, pad64=(0, 1))
    assert torch.equal(o2, outs[0]) and float((s - xs[1].float().sum(0)).abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,C", [(77, 320), (16384, 320), (4096, 1280), (100, 8), (1000, 136), (64, 64), (2, 5120)])
def test_transpose2d_many_fused_colsum(dev, dtype, R, C):
    """the bias gradient computed by the transpose launch (ur_transpose_desc.colsum): == fp64 column sums of the same
    rounded values to fp32 accuracy, bitwise reproducible, transposes unchanged, counters left at zero."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(R + C)
    y = (torch.randn(R, C, generator=g) + 0.25).to(dtype).to(dev)
    other = torch.randn(40, 72, generator=g).to(dtype).to(dev)
    (ot, yt), s = bw.transpose2d_many([other, y], colsum_of=1)
    assert torch.equal(yt, bw.transpose2d(y)) and torch.equal(ot, bw.transpose2d(other))
    want = y.double().sum(0)
    assert s.dtype == torch.float32 and s.shape == (C,)
    assert float((s.double() - want).abs().max()) <= 2e-6 * float(y.double().abs().sum(0).max())
    (_, _), s2 = bw.transpose2d_many([other, y], colsum_of=1)
    assert torch.equal(s, s2)
    assert int(bw._colsum_counter(y.device).abs().sum()) == 0
    assert float((bw.colsum(y).double() - want).abs().max()) <= 2e-6 * float(y.double().abs().sum(0).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2, 40, 128, 128), (1, 2, 40, 100, 77), (1, 2, 80, 64, 64), (1, 2, 160, 96, 40)])
def test_attention_forward_lse(dev, dtype, shape):
    """ur_attn_desc.lse: the forward kernels' row log-sum-exp (log2 units) against fp32 logsumexp of the scaled scores."""
    from uni_renderer_amd import backward as bw, ops
    B, H, d, Tq, Tk = shape
    C = H * d
    q, k, v = _rand((B, Tq, C), dtype, dev, 1), _rand((B, Tk, C), dtype, dev, 2), _rand((B, Tk, C), dtype, dev, 3)
    vt = bw._pad_rows64(bw.transpose2d(v))
    lse = torch.full((B * H, Tq), float("nan"), dtype=torch.float32, device=dev)
    o = ops.attention(q, k, vt, B=B, H=H, Tq=Tq, Tk=Tk, d=d, ldq=C, ldk=C, lse=lse)
    o0 = ops.attention(q, k, vt, B=B, H=H, Tq=Tq, Tk=Tk, d=d, ldq=C, ldk=C)
    assert torch.equal(o, o0)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float().view(B, Tq, H, d), k.float().view(B, Tk, H, d)) * d ** -0.5
    ref = torch.logsumexp(s, dim=-1).reshape(B * H, Tq) * 1.4426950408889634
    err = float((lse - ref).abs().max())
    print({"attention_forward_lse": str(dtype), "shape": shape, "max_abs_err_log2": err})
    assert err < (2e-3 if dtype == torch.float16 else 1.5e-2)

@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2, 40, 128, 77), (1, 8, 40, 1024, 77), (4, 8, 40, 4096, 77), (1, 2, 80, 64, 77),
                                   (1, 2, 160, 64, 40), (1, 2, 40, 128, 200), (1, 4, 40, 256, 128)])
@pytest.mark.parametrize("direct", [None, 0, 1000])
def test_flash_attention_backward_cross(dev, dtype, shape, direct, monkeypatch):
    """The flash backward with fewer (padded, masked) keys than queries -- the 77-key cross-attention -- incl. shapes whose
    dk / dv kernel splits the queries and folds fp32 partial sums; through autograd_ops.Attention (forward log-sum-exp
    handed over) and directly (log-sum-exp pass in the dq kernel), against fp32 SDPA autograd."""
    from uni_renderer_amd import backward as bw
    from uni_renderer_amd import autograd_ops as A
    if direct is not None:
        monkeypatch.setattr(bw, "FLASH_DIRECT_MIN_D", direct)
    B, H, d, Tq, Tk = shape
    C = H * d
    q, do = _rand((B, Tq, C), dtype, dev, 1), _rand((B, Tq, C), dtype, dev, 4)
    k, v = _rand((B, Tk, C), dtype, dev, 2), _rand((B, Tk, C), dtype, dev, 3)
    qr, kr, vr = (t.float().cpu().requires_grad_() for t in (q, k, v))
    o_ref = F.scaled_dot_product_attention(qr.view(B, Tq, H, d).transpose(1, 2), kr.view(B, Tk, H, d).transpose(1, 2),
                                           vr.view(B, Tk, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, Tq, C)
    (o_ref * do.float().cpu()).sum().backward()
    lib = bw._lib.load()
    assert lib.ur_attention_backward_supported(Tq, Tk, d)
    q_, k_, v_ = (t.clone().requires_grad_() for t in (q, k, v))
    o = A.Attention.apply(q_, k_, v_, H)
    o.backward(do)
    direct = bw.attention_backward(q, k, v, do, H, o=o.detach())  # no stats: the dq kernel computes the log-sum-exp
    tol = TOL[dtype] * 2
    errs = [rel_l2(q_.grad, qr.grad), rel_l2(k_.grad, kr.grad), rel_l2(v_.grad, vr.grad)]
    errs2 = [rel_l2(a, b.grad) for a, b in zip(direct, (qr, kr, vr))]
    G = lib.ur_attention_backward_splits(B * H, Tq, (Tk + 63) // 64 * 64, (d + 31) // 32 * 32)
    print({"flash_attention_backward_cross": str(dtype), "shape": shape, "query_splits": G, "rel_l2_dq_dk_dv": errs,
           "own_lse": errs2})
    assert rel_l2(o, o_ref) < TOL[dtype] and max(errs) < tol and max(errs2) < tol, (errs, errs2)

def test_cast_many(dev):
    """ur_cast_multi: > 128 tensors, odd sizes and unaligned views, both directions, against Tensor.to."""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(9)
    base = torch.randn(100000, generator=g).to(dev)
    srcs = [torch.randn(n, generator=g).to(dev) for n in (1, 7, 8, 4096, 8192, 8193, 100003)] * 20 + [base[1:50002], base[3:11]]
    for dt in (torch.bfloat16, torch.float16):
        outs = bw.cast_many(srcs, dt)
        for a, b in zip(srcs, outs):
            assert b.dtype == dt and torch.equal(b, a.to(dt))
        back = bw.cast_many(outs, torch.float32)
        for a, b in zip(outs, back):
            assert b.dtype == torch.float32 and torch.equal(b, a.float())


def test_gradient_sums_of_squares_from_the_writing_kernels(dev):
    """ur_cast_multi_sumsq / ur_unpack_conv_weight_grad_sumsq: the per-workgroup partial sums add up to the sum of squares of
    exactly what was written (fp64 reference), the outputs are unchanged, and the result is reproducible bit for bit."""
    import ctypes as C

    from uni_renderer_amd import _lib, autograd_ops as A, backward as bw
    g = torch.Generator().manual_seed(13)
    srcs = [(torch.randn(n, generator=g) * 3).to(torch.bfloat16).to(dev) for n in (1, 7, 8193, 100003, 320 * 1280)] * 30
    outs, part = bw.cast_many(srcs, torch.float32, sumsq=True)
    want = sum((s.double() ** 2).sum() for s in srcs)
    assert all(torch.equal(o, s.float()) for o, s in zip(outs, srcs))
    assert abs(float(part.double().sum()) - float(want)) <= 1e-6 * float(want)
    _, part2 = bw.cast_many(srcs, torch.float32, sumsq=True)
    assert torch.equal(part, part2)
    lib = _lib.load()
    for co, ci, cp in ((320, 320, 320), (4, 28, 64), (1280, 2560, 2560)):
        dwp = (torch.randn(co, 9 * cp, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        ref = torch.empty(co, ci, 3, 3, dtype=torch.float32, device=dev)
        bw.check(lib.ur_unpack_conv_weight_grad(dwp.data_ptr(), dwp.stride(0), ref.data_ptr(), co, ci, cp, bw.DT[dwp.dtype], None),
                 "ur_unpack_conv_weight_grad")
        out = torch.empty_like(ref)
        ps = torch.empty(int(lib.ur_unpack_conv_weight_grad_blocks(co, ci)), dtype=torch.float32, device=dev)
        bw.check(lib.ur_unpack_conv_weight_grad_sumsq(dwp.data_ptr(), dwp.stride(0), out.data_ptr(), co, ci, cp, ps.data_ptr(),
                                                      bw.DT[dwp.dtype], None), "ur_unpack_conv_weight_grad_sumsq")
        torch.cuda.synchronize()
        w = float((ref.double() ** 2).sum())
        assert torch.equal(out, ref) and abs(float(ps.double().sum()) - w) <= 1e-6 * w


def test_clipping_norm_from_the_backward_matches_the_sweep(dev):
    """train_step's clipping norm assembled from the per-launch sums of squares (backward.GradSquares) equals
    torch's norm over the same gradients; a parameter with two contributions (one conv weight packed twice) makes the
    registry unusable, so the step falls back to the sweep."""
    import torch.nn as nn

    from uni_renderer_amd import autograd_ops as A, backward as bw
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 64, 3, padding=1).to(dev)
    lin = nn.Linear(64, 128).to(dev)
    x = torch.randn(2, 16, 16, 64, device=dev, dtype=torch.bfloat16)

    def run(twice):
        for p in list(conv.parameters()) + list(lin.parameters()):
            p.grad = None
        bw.grad_squares.begin()
        (wl,) = A.CastParams.apply(torch.bfloat16, lin.weight)
        h = A.conv3x3(x, A.pack_conv_weight(conv.weight, torch.bfloat16), conv.bias)
        if twice:
            h = A.conv3x3(h, A.pack_conv_weight(conv.weight, torch.bfloat16), conv.bias)
        y = A.linear(h, wl, lin.bias)
        (y.float() ** 2).mean().backward()

    run(False)
    gs = bw.grad_squares
    assert gs.usable() and set(gs.count) == {id(conv.weight), id(lin.weight)}
    fused = torch.cat(gs.parts).double().sum() + sum((p.grad.double() ** 2).sum() for p in (conv.bias, lin.bias))
    ref = sum((p.grad.double() ** 2).sum() for p in list(conv.parameters()) + list(lin.parameters()))
    assert abs(float(fused) - float(ref)) <= 1e-6 * float(ref)
    run(True)
    assert not bw.grad_squares.usable() and bw.grad_squares.count[id(conv.weight)] == 2


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32])
@pytest.mark.parametrize("M,N,rpg", [(16384, 320, 0), (600, 136, 0), (8192, 1280, 4096), (77 * 4, 768, 0), (64, 8, 0), (100000, 64, 0)])
def test_colsum_one_launch_equals_two_launches(dev, dtype, M, N, rpg, monkeypatch):
    """the last-arriver fold inside ur_colsum_fused gives the bits of partial sums + colsum_fold_kernel, run after run, and
    leaves its counters at zero"""
    from uni_renderer_amd import backward as bw
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, N, generator=g) + 0.1).to(dtype).to(dev)
    monkeypatch.setattr(bw, "COLSUM_ONE_LAUNCH", False)
    two = bw.colsum(x, rows_per_group=rpg)
    monkeypatch.setattr(bw, "COLSUM_ONE_LAUNCH", True)
    for _ in range(20):
        one = bw.colsum(x, rows_per_group=rpg)
        assert torch.equal(one, two)
    assert int(bw._colsum_counter(x.device).abs().sum()) == 0
    want = x.double().sum(0) if rpg == 0 else x.double().view(-1, rpg, N).sum(1)
    assert float((one.double() - want).abs().max()) <= 2e-6 * float(x.double().abs().sum(0).max())


def test_packed_casts_concatenate_without_a_copy(dev):
    """CastParams lays its copies out back to back; cat_adjacent over neighbours is the same matrix as torch.cat, shares their
    storage, and routes every row slice of the gradient back to its parameter."""
    from uni_renderer_amd import autograd_ops as A
    torch.manual_seed(5)
    ws = [torch.randn(n, 64, device=dev, requires_grad=True) for n in (32, 48, 16, 8)]
    cs = A.CastParams.apply(torch.bfloat16, *ws)
    cat = A.cat_adjacent(cs[:3])
    assert cat.shape == (96, 64) and cat.data_ptr() == cs[0].data_ptr()
    assert torch.equal(cat, torch.cat([w.detach().to(torch.bfloat16) for w in ws[:3]], 0))
    up = torch.randn(96, 64, device=dev).to(torch.bfloat16)
    (cat.float() * up.float()).sum().backward()
    for w, g in zip(ws[:3], torch.split(up.float(), [32, 48, 16], 0)):
        assert torch.equal(w.grad, g)
    assert ws[3].grad is None
    other = A.cat_adjacent([cs[0], cs[2]])   # not neighbours: falls back to a copy
    assert other.data_ptr() != cs[0].data_ptr() and torch.equal(other, torch.cat([cs[0], cs[2]], 0))


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_skip_node(dev, dtype):
    """LayerNormSkip: x + f(LN(x)) with the residual's gradient added inside the LayerNorm backward kernel == the two-node
    form (LayerNorm + autograd's accumulation) up to the rounding of that one addition, and == fp32 autograd."""
    from uni_renderer_amd import autograd_ops as A
    C = 640
    x0 = _rand((3, 50, C), dtype, dev, 1)
    g_ = (1 + 0.1 * torch.randn(C)).to(dev)
    b_ = (0.1 * torch.randn(C)).to(dev)
    w = _rand((C, C), dtype, dev, 2, C ** -0.5)
    up = _rand((3, 50, C), dtype, dev, 3)

    def run(skipnode):
        x = x0.clone().requires_grad_()
        g, b = g_.clone().requires_grad_(), b_.clone().requires_grad_()
        if skipnode:
            xs, xn = A.LayerNormSkip.apply(x, g, b, 1e-5)
        else:
            xs, xn = x, A.LayerNorm.apply(x, g, b, 1e-5)
        y = A.linear(xn, w, None, res=xs)
        y.backward(up)
        return y.detach(), x.grad, g.grad, b.grad

    y1, dx1, dg1, db1 = run(True)
    y0, dx0, dg0, db0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)
    assert rel_l2(dx1, dx0.float()) < (1e-3 if dtype == torch.float16 else 6e-3)
    xr = x0.float().cpu().requires_grad_()
    yr = F.layer_norm(xr, (C,), g_.cpu(), b_.cpu(), 1e-5) @ w.float().cpu().t() + xr
    yr.backward(up.float().cpu())
    assert rel_l2(dx1, xr.grad) < TOL[dtype]
