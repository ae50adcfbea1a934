"""Weight / bias gradients without transposed operand copies (``ur_wgrad``, csrc/wgrad.hip) against an fp32 reference on the
rounded inputs: dw = dy^T . x for a linear layer, the packed [N][(ky, kx, c)] weight gradient of a 3x3 conv (implicit
im2col, stride 1 | 2), db = column sums of dy.  Reference: the autograd of F.linear / F.conv2d under
train/train.py:1416 (``accelerator.backward``); the shapes are the layers of models/unet_2d_blocks.py at cfg 4."""
import pytest
import torch
import torch.nn.functional as F

from uni_renderer_amd import backward as B_

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}


def _mk(shape, dt, s=1.0):
    return (torch.randn(*shape, device="cuda") * s).to(dt)


def _close(got, ref, dt, what):
    err = (got.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    assert err <= TOL[dt], f"{what}: {err:.3e}"
    return err


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("P,N,K,splits", [(4096, 320, 320, 0), (308, 320, 768, 1), (308, 640, 768, 3), (5, 1280, 320, 1),
                                          (1000, 8, 8, 0), (16384, 320, 1280, 0), (33, 64, 2560, 2), (2048, 1280, 5120, 0)])
def test_linear_weight_and_bias_gradients(dt, tile, P, N, K, splits):
    torch.manual_seed(P + N + K)
    dy, x = _mk((P, N), dt), _mk((P, K), dt)
    dw, db = B_.wgrad(dy, x, True, tile=tile, splits=splits)
    assert dw.shape == (N, K) and db.shape == (N,) and db.dtype == torch.float32
    _close(dw, dy.float().t() @ x.float(), dt, "dw")          # a transposed result would not pass: dy and x are independent
    ref_db = dy.float().sum(0)
    assert (db - ref_db).abs().max().item() <= 2e-4 * max(1.0, dy.float().abs().sum(0).max().item())
    dw2, db2 = B_.wgrad(dy, x, True, tile=tile, splits=splits)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)      # fixed-order reductions
    dw3, none = B_.wgrad(dy, x, False, tile=tile, splits=splits)
    assert none is None and torch.equal(dw, dw3)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_linear_gradient_from_a_column_slice(dt):
    """dy as a column slice of a wider matrix (the k part of a fused q | k | v gradient) and x with a padded row stride."""
    torch.manual_seed(3)
    wide, xw = _mk((700, 960), dt), _mk((700, 384), dt)
    dy, x = wide[:, 320:640], xw[:, :320]
    dw, db = B_.wgrad(dy, x, True)
    _close(dw, dy.float().t() @ x.float(), dt, "dw")
    assert (db - dy.float().sum(0)).abs().max().item() <= 2e-3


def _conv_ref(x, dy, stride):
    """fp32 weight gradient of F.conv2d(x, w, padding=1, stride) in the packed layout [N][(ky, kx, c)]."""
    Bn, H, W, Cc = x.shape
    N = dy.shape[-1]
    w = torch.zeros(N, Cc, 3, 3, device="cuda", requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1, stride=stride)
    (g,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
    return g.permute(0, 2, 3, 1).reshape(N, 9 * Cc)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("Bn,H,W,Cc,N,stride,splits", [(2, 16, 16, 64, 64, 1, 0), (1, 32, 32, 320, 320, 1, 0), (2, 8, 8, 640, 1280, 1, 2),
                                                       (1, 16, 32, 128, 320, 1, 3), (2, 32, 32, 320, 320, 2, 0), (3, 16, 16, 192, 64, 2, 1),
                                                       (4, 64, 64, 320, 320, 1, 0), (1, 8, 8, 2560, 1280, 1, 0)])
def test_conv3x3_weight_and_bias_gradients(dt, tile, Bn, H, W, Cc, N, stride, splits):
    torch.manual_seed(Bn + H + Cc + N + stride)
    Ho, Wo = H // stride, W // stride
    x, dy = _mk((Bn, H, W, Cc), dt), _mk((Bn, Ho, Wo, N), dt, 0.5)
    dw, db = B_.wgrad(dy.reshape(-1, N), x, True, conv=(Ho, Wo, stride), tile=tile, splits=splits)
    assert dw.shape == (N, 9 * Cc)
    _close(dw, _conv_ref(x, dy, stride), dt, "conv dw")
    assert (db - dy.float().sum((0, 1, 2))).abs().max().item() <= 2e-4 * max(1.0, dy.float().abs().sum((0, 1, 2)).max().item())
    dw2, _ = B_.wgrad(dy.reshape(-1, N), x, True, conv=(Ho, Wo, stride), tile=tile, splits=splits)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_backward_entry_points_agree_with_the_transposed_operand_path(dt, monkeypatch):
    """linear_backward / conv3x3_backward with and without ur_wgrad: same dx bits, dw / db within the storage rounding."""
    torch.manual_seed(11)
    x, w, dy = _mk((2, 300, 320), dt), _mk((640, 320), dt, 0.05), _mk((2, 300, 640), dt)
    xc, wc, dyc = _mk((2, 16, 16, 320), dt), _mk((640, 9 * 320), dt, 0.02), _mk((2, 16, 16, 640), dt)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(B_, "WGRAD", flag)
        outs[flag] = B_.linear_backward(x, w, dy) + B_.conv3x3_backward(xc, wc, dyc)
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][3], outs[False][3])
    for i in (1, 4):
        _close(outs[True][i], outs[False][i].float(), dt, "dw vs transposed path")
    for i in (2, 5):
        assert (outs[True][i] - outs[False][i]).abs().max().item() <= 1e-3 * max(1.0, outs[False][i].abs().max().item())


def test_descriptor_validation():
    dy, x = _mk((64, 20), torch.float16), _mk((64, 64), torch.float16)
    assert not B_.wgrad_ok(dy, x)                      # N % 8
    with pytest.raises(RuntimeError):
        B_.wgrad(dy, x)
    xc = _mk((1, 12, 12, 64), torch.float16)
    assert not B_.wgrad_ok(_mk((144, 64), torch.float16), xc, (12, 12))   # output size not a power of two
    with pytest.raises(RuntimeError):
        B_.wgrad(_mk((144, 64), torch.float16), xc, conv=(12, 12, 1))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("P,N,K,n,tile,splits", [(4096, 320, 320, 7, 1, 1), (1024, 640, 640, 64, 3, 2), (308, 320, 768, 5, 0, 0),
                                                 (16384, 320, 320, 12, 1, 4), (50, 8, 2560, 3, 2, 1), (2048, 1280, 320, 9, 5, 0)])
def test_grouped_problems_are_the_single_launches(dt, P, N, K, n, tile, splits):
    """ur_wgrad_group: n equally shaped problems in one launch give the bits of n single launches with the same tile / slices;
    a problem without a bias gradient in the middle of the group."""
    torch.manual_seed(P + N + n)
    probs = [(_mk((P, N), dt), _mk((P, K), dt)) for _ in range(n)]
    items = [(dy, x, torch.empty(N, K, dtype=dt, device="cuda"),
              None if i == 1 else torch.empty(N, dtype=torch.float32, device="cuda")) for i, (dy, x) in enumerate(probs)]
    trace = {}
    B_.wgrad_group(items, tile=tile, splits=splits, trace=trace)
    assert trace == {f"{P},{N},{K},1,0@{n}": 1}
    for i, (dy, x, dw, db) in enumerate(items):
        _close(dw, dy.float().t() @ x.float(), dt, f"dw[{i}]")
        if tile and splits:
            dw1, db1 = B_.wgrad(dy, x, db is not None, tile=tile, splits=splits)
            assert torch.equal(dw, dw1) and (db is None or torch.equal(db, db1))
        elif db is not None:
            assert (db - dy.float().sum(0)).abs().max().item() <= 2e-4 * max(1.0, dy.float().abs().sum(0).max().item())


def test_queue_groups_by_shape_and_flushes_on_a_repeated_weight():
    dt = torch.bfloat16
    q = B_.WgradQueue()
    q.trace = {}
    a = [(_mk((256, 64), dt), _mk((256, 128), dt)) for _ in range(3)]
    b = [(_mk((512, 64), dt), _mk((512, 128), dt)) for _ in range(2)]
    outs = [q.add(dy, x, 1000 + i, True) for i, (dy, x) in enumerate(a + b)]
    assert all(o is not None for o in outs) and len(q.items) == 5
    assert q.add(a[0][0], a[0][1], 1000, True) is None and not q.items       # same weight again: flushed, caller computes at once
    assert q.trace == {"256,64,128,1,0@3": 1, "512,64,128,1,0@2": 1}
    for (dy, x), (dw, db) in zip(a + b, outs):
        _close(dw, dy.float().t() @ x.float(), dt, "queued dw")
        assert (db - dy.float().sum(0)).abs().max().item() <= 1e-2


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Bn,H,Cc,N,stride,n,tile,splits", [(2, 16, 64, 64, 1, 5, 3, 1), (1, 32, 320, 320, 1, 3, 2, 2), (2, 16, 128, 320, 2, 4, 1, 1),
                                                            (4, 8, 1280, 1280, 1, 2, 5, 1)])
def test_grouped_conv_problems_are_the_single_launches(dt, Bn, H, Cc, N, stride, n, tile, splits):
    torch.manual_seed(Bn + H + Cc + n)
    Ho = H // stride
    probs = [(_mk((Bn, Ho, Ho, N), dt, 0.5), _mk((Bn, H, H, Cc), dt)) for _ in range(n)]
    items = [(dy.reshape(-1, N), x, torch.empty(N, 9 * Cc, dtype=dt, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda"))
             for dy, x in probs]
    B_.wgrad_group(items, tile=tile, splits=splits, conv=(Ho, Ho, stride))
    for (dy, x), (_, _, dw, db) in zip(probs, items):
        _close(dw, _conv_ref(x, dy, stride), dt, "grouped conv dw")
        dw1, db1 = B_.wgrad(dy.reshape(-1, N), x, True, conv=(Ho, Ho, stride), tile=tile, splits=splits)
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
