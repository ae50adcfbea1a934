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
