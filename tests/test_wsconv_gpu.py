"""Weight-streaming 3x3 conv (csrc/wsconv.hip, tile UR_TILE_WS320 of ur_igemm) against an fp32 reference and against the
LDS-tiled implicit GEMM on the same packed weights: resnet conv1 / conv2 shapes of the UNets (reference:
models/unet_2d_blocks.py ResnetBlock2D via SURVEY.md rows a11 - a13), with time-embedding row add, (hi, lo) residual,
1x1 shortcut tail, grouped streams and split-K."""
import pytest
import torch
import torch.nn.functional as F

from uni_renderer_amd import ops
from uni_renderer_amd.layers import pack_conv3x3, pack_matrix

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ops.wsconv_built(), reason="weight-streaming conv tiles are an opt-in build (make WSCONV=1): "
                                 "measured at parity / slower in the step, not part of the product library (DESIGN.md section 4)")]


def _ref(x, wt, b, rowadd=None, res=None, tail=None, wtail=None, out_scale=1.0):
    """fp32 NCHW reference on the ROUNDED inputs."""
    y = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), b.float(), padding=1)
    if tail is not None:
        t = torch.cat([u.float() for u in tail if u is not None], -1).permute(0, 3, 1, 2)
        y = y + F.conv2d(t, wtail.float()[:, :, None, None])
    y = y.permute(0, 2, 3, 1)
    if rowadd is not None:
        y = y + rowadd.float()[:, None, None, :]
    if res is not None:
        y = y + res.float() + (ops.lo_float(res.lo) if ops.lo_of(res) is not None else 0)
    return y * out_scale


def _mk(shape, dt, s=1.0):
    return (torch.randn(*shape, device="cuda") * s).to(dt)


@pytest.mark.parametrize("wtile", [47, 48])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,N,splitk", [(2, 16, 16, 320, 320, 1), (1, 16, 8, 640, 320, 1), (2, 16, 16, 320, 640, 2),
                                                (1, 32, 32, 960, 320, 3), (1, 16, 16, 1280, 640, 4), (3, 8, 16, 320, 320, 1)])
def test_wsconv_vs_fp32_and_igemm(dt, B, H, W, Cin, N, splitk, wtile):
    torch.manual_seed(B * 1000 + Cin + N + splitk)
    x = _mk((B, H, W, Cin), dt)
    wt = _mk((N, Cin, 3, 3), dt, (9 * Cin) ** -0.5)
    b = torch.randn(N, device="cuda")
    ra = _mk((B, N), dt)
    cb = ops.conv_cblock(Cin)
    w = pack_conv3x3(wt, dt, cblock=cb)
    ws = ops.wsconv_images(w, N)
    y = ops.conv3x3(x, w, b, rowadd=ra, cblock=cb, ws=ws, splitk=splitk, tile=wtile)
    y0 = ops.conv3x3(x, w, b, rowadd=ra, cblock=cb)
    ref = _ref(x, wt, b, rowadd=ra)
    tol = 2e-3 if dt == torch.float16 else 1.5e-2
    assert (y.float() - ref).abs().max() <= tol * ref.abs().max()
    assert (y.float() - y0.float()).abs().max() <= tol * ref.abs().max()


@pytest.mark.parametrize("wtile", [47, 48])
@pytest.mark.parametrize("splitk", [1, 2])
def test_wsconv_residual_hilo_and_scale(splitk, wtile):
    torch.manual_seed(5)
    dt, B, H, W, C = torch.float16, 2, 16, 16, 320
    x = _mk((B, H, W, C), dt)
    wt = _mk((C, C, 3, 3), dt, (9 * C) ** -0.5)
    b = torch.randn(C, device="cuda")
    resf = torch.randn(B, H, W, C, device="cuda")
    res = resf.to(dt)
    res.lo = ops.lo_encode(resf - res.float(), dt)
    w = pack_conv3x3(wt, dt)
    y = ops.conv3x3(x, w, b, res=res, out_scale=0.5, hilo=True, ws=ops.wsconv_images(w), splitk=splitk, tile=wtile)
    ref = _ref(x, wt, b, res=res, out_scale=0.5)
    got = y.float() + ops.lo_float(y.lo)
    assert (got - ref).abs().max() <= 3e-4 * ref.abs().max()      # the (hi, lo) pair carries ~2^-14 relative
    assert (y.float() - ref).abs().max() <= 1e-3 * ref.abs().max()


@pytest.mark.parametrize("wtile", [47, 48])
@pytest.mark.parametrize("Ct0,Ct1,splitk", [(320, 0, 1), (640, 320, 1), (320, 320, 2)])
def test_wsconv_shortcut_tail(Ct0, Ct1, splitk, wtile):
    torch.manual_seed(7 + Ct0 + Ct1)
    dt, B, H, W, C, N = torch.float16, 2, 16, 16, 640, 640
    x = _mk((B, H, W, C), dt)
    t0 = _mk((B, H, W, Ct0), dt)
    t1 = _mk((B, H, W, Ct1), dt) if Ct1 else None
    wt = _mk((N, C, 3, 3), dt, (9 * C) ** -0.5)
    wtail = _mk((N, Ct0 + Ct1), dt, (Ct0 + Ct1) ** -0.5)
    b = torch.randn(N, device="cuda")
    cb = ops.conv_cblock(C)
    w = torch.cat([pack_conv3x3(wt, dt, cblock=cb), pack_matrix(wtail, dt)], 1).contiguous()
    y = ops.conv3x3(x, w, b, tail=(t0, t1), cblock=cb, ws=ops.wsconv_images(w), splitk=splitk, hilo=True, tile=wtile)
    y0 = ops.conv3x3(x, w, b, tail=(t0, t1), cblock=cb, hilo=True)
    ref = _ref(x, wt, b, tail=(t0, t1), wtail=wtail)
    assert (y.float() - ref).abs().max() <= 2e-3 * ref.abs().max()
    assert (y.float() - y0.float()).abs().max() <= 2e-3 * ref.abs().max()


@pytest.mark.parametrize("wtile", [47, 48])
@pytest.mark.parametrize("splitk", [1, 2])
def test_wsconv_two_streams(splitk, wtile):
    """grouped launch: stream-major batch, per-stream weights / bias / time-embedding rows"""
    torch.manual_seed(11)
    dt, S, B, H, W, C, N = torch.float16, 2, 2, 16, 16, 640, 320
    x = _mk((S * B, H, W, C), dt)
    wts = [_mk((N, C, 3, 3), dt, (9 * C) ** -0.5) for _ in range(S)]
    bs = torch.randn(S, N, device="cuda")
    ra = _mk((S * B, N), dt)
    cb = ops.conv_cblock(C)
    w = torch.stack([pack_conv3x3(t, dt, cblock=cb) for t in wts])
    ws = torch.stack([ops.wsconv_images(w[i]) for i in range(S)])
    y = ops.conv3x3(x, w, bs, rowadd=ra, streams=S, cblock=cb, ws=ws, splitk=splitk, tile=wtile)
    for s_ in range(S):
        ref = _ref(x[s_ * B:(s_ + 1) * B], wts[s_], bs[s_], rowadd=ra[s_ * B:(s_ + 1) * B])
        assert (y[s_ * B:(s_ + 1) * B].float() - ref).abs().max() <= 2e-3 * ref.abs().max()


def test_wsconv_unsupported_shapes_fall_back():
    dt = torch.float16
    x = _mk((1, 8, 8, 320), dt)   # 64 pixels per sample: not a multiple of 128
    assert not ops.wsconv_ok(x, 320)
    assert not ops.wsconv_ok(_mk((1, 16, 16, 256), dt), 320)
    assert ops.wsconv_ok(_mk((1, 16, 16, 320), dt), 640)
