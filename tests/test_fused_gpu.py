"""The grouped ("two streams, one launch") executor must compute the same step as the module-by-module path and
as the CPU oracle: inverse direction, rendering direction, a conditioning scale != 1, fp16 and bf16."""
import pytest
import torch

from conftest import rel_l2
from util_models import O, build_product_from_oracle, product_step

pytestmark = pytest.mark.gpu


def _setup(dev, dtype, seed=31, B=2):
    unet_o, enc_o, dec_o = O.build_triplet(O.TINY_CONFIG, seed=seed)
    unet, enc, dec = build_product_from_oracle(unet_o, enc_o, dec_o, dtype, dev)
    x, c, ehs, ti, ta = O.make_inputs(B, 16, 64, seed=seed + 1)
    return (unet_o, enc_o, dec_o), (unet, enc, dec), (x, c, ehs, ti, ta)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.5e-2)])
def test_grouped_step_matches_modules_and_oracle(dev, dtype, tol):
    from uni_renderer_amd.fused import GroupedDualStreamStep

    oracle, (unet, enc, dec), (x, c, ehs, ti, ta) = _setup(dev, dtype)
    ref = O.dual_stream_step(*oracle, x, c, ehs, ti, ta)
    g = [t.to(dev) for t in (x, c, ehs, ti, ta)]
    with torch.no_grad():
        mod = product_step(unet, enc, dec, *g)
        out = GroupedDualStreamStep(unet, enc, dec)(*g)
    assert out["img_pred"].shape == (2, 4, 16, 16) and out["attr_pred"].shape == (2, 28, 16, 16)
    for k in ("img_pred", "attr_pred"):
        assert rel_l2(out[k], ref[k]) < tol, (k, rel_l2(out[k], ref[k]))
        assert rel_l2(out[k], mod[k]) < tol, (k, rel_l2(out[k], mod[k]))


def test_grouped_rendering_direction_and_scale(dev):
    from uni_renderer_amd.fused import GroupedDualStreamStep

    oracle, (unet, enc, dec), (x, c, ehs, ti, ta) = _setup(dev, torch.float16, seed=41, B=3)
    unet_o, enc_o, dec_o = oracle
    g = [t.to(dev) for t in (x, c, ehs, ti, ta)]
    step = GroupedDualStreamStep(unet, enc, dec)
    with torch.no_grad():
        out = step(*g, run_decoder=False)
    ref = O.dual_stream_step(*oracle, x, c, ehs, ti, ta, run_decoder=False)
    assert "attr_pred" not in out and rel_l2(out["img_pred"], ref["img_pred"]) < 3e-3
    # conditioning_scale (controlnet.py:1773-1775): enc residuals scaled before they enter the unet
    with torch.no_grad():
        res, mid, raw_enc, raw_mid_enc = enc_o(x, ta, ehs, controlnet_cond=c, conditioning_scale=0.5)
        ref2 = unet_o(x, ti, ehs, res, mid)[0]
        out2 = step(*g, run_decoder=True, conditioning_scale=0.5)
    assert rel_l2(out2["img_pred"], ref2) < 3e-3


def test_grouped_single_timestep_and_single_prompt_broadcast(dev):
    from uni_renderer_amd.fused import GroupedDualStreamStep

    oracle, (unet, enc, dec), (x, c, ehs, ti, ta) = _setup(dev, torch.float16, seed=51)
    ref = O.dual_stream_step(*oracle, x, c, ehs[:1].expand(2, -1, -1), torch.tensor([7, 7]), torch.tensor([500, 500]))
    with torch.no_grad():
        out = GroupedDualStreamStep(unet, enc, dec)(x.to(dev), c.to(dev), ehs[:1].to(dev), 7, torch.tensor(500, device=dev))
    assert rel_l2(out["img_pred"], ref["img_pred"]) < 3e-3 and rel_l2(out["attr_pred"], ref["attr_pred"]) < 3e-3


@pytest.mark.parametrize("hw", [(12, 12), (16, 24), (12, 20), (9, 14)])
def test_odd_and_non_square_latents(dev, hw):
    """Latent sides that are not multiples of 8 make the reference forward ``upsample_size`` (controlnet.py:869-883,
    1129-1130: F.interpolate(size=skip size) instead of x2); non-square maps exercise every H != W index path.
    Module path and grouped executor against the CPU oracle."""
    from uni_renderer_amd.fused import GroupedDualStreamStep

    oracle = O.build_triplet(O.TINY_CONFIG, seed=61)
    unet, enc, dec = build_product_from_oracle(*oracle, torch.float16, dev)
    g = torch.Generator().manual_seed(62)
    H, W = hw
    x, c = torch.randn(2, 4, H, W, generator=g), torch.randn(2, 28, H, W, generator=g)
    ehs = torch.randn(2, 77, 64, generator=g) * 0.5
    ti, ta = torch.tensor([3, 700]), torch.tensor([999, 0])
    ref = O.dual_stream_step(*oracle, x, c, ehs, ti, ta)
    d = [t.to(dev) for t in (x, c, ehs, ti, ta)]
    with torch.no_grad():
        mod = product_step(unet, enc, dec, *d)
        grp = GroupedDualStreamStep(unet, enc, dec)(*d)
    for out in (mod, grp):
        assert out["img_pred"].shape == (2, 4, H, W) and out["attr_pred"].shape == (2, 28, H, W)
        for k in ("img_pred", "attr_pred"):
            assert rel_l2(out[k], ref[k]) < 3e-3, (hw, k, rel_l2(out[k], ref[k]))


@pytest.mark.parametrize("hw", [(16, 16), (12, 20), (9, 14)])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.5e-2)])
def test_hoisted_steps_match_the_oracle_step(dev, hw, dtype, tol):
    """hoist.HoistedSamplingStep op-level (no sampler around it): prologue on the fixed inputs + one step on the evolving ones
    must be the oracle's enc -> unet -> dec step -- inverse direction (UNet down + mid and the decoder's exchange products from
    the prologue, UNet up never run) and rendering direction (encoder from the prologue, conditioning scale 0.5) -- also on
    latent sides that are not multiples of 8 / non-square maps, with per-sample timesteps, fp16 and bf16; a second step with
    OTHER evolving inputs reuses the prologue."""
    from uni_renderer_amd.hoist import HoistedSamplingStep

    oracle = O.build_triplet(O.TINY_CONFIG, seed=71)
    unet_o, enc_o, dec_o = oracle
    unet, enc, dec = build_product_from_oracle(*oracle, dtype, dev)
    g = torch.Generator().manual_seed(72)
    H, W = hw
    x, c = torch.randn(2, 4, H, W, generator=g), torch.randn(2, 28, H, W, generator=g)
    c2, x2 = torch.randn(2, 28, H, W, generator=g), torch.randn(2, 4, H, W, generator=g)
    ehs = torch.randn(2, 77, 64, generator=g) * 0.5
    ti, ta, tb = torch.tensor([0, 0]), torch.tensor([999, 340]), torch.tensor([120, 7])
    d = lambda t: t.to(dev)
    inv = HoistedSamplingStep(unet, enc, dec, "inverse")
    inv.prologue(d(x), d(ehs), d(ti))
    for cond, t in ((c, ta), (c2, tb)):
        out = inv.step(d(cond), d(t))["attr_pred"]
        ref = O.dual_stream_step(*oracle, x, cond, ehs, ti, t)["attr_pred"]
        assert out.shape == (2, 28, H, W) and rel_l2(out, ref) < tol, (hw, rel_l2(out, ref))
    ren = HoistedSamplingStep(unet, enc, dec, "render", conditioning_scale=0.5)
    ren.prologue(d(c), d(ehs), d(ti))  # clean attributes, t_attr = 0
    for xt, t in ((x, ta), (x2, tb)):
        out = ren.step(d(xt), d(t))["img_pred"]
        with torch.no_grad():
            res, mid, _, _ = enc_o(xt, ti, ehs, controlnet_cond=c, conditioning_scale=0.5)
            ref = unet_o(xt, t, ehs, res, mid)[0]
        assert out.shape == (2, 4, H, W) and rel_l2(out, ref) < tol, (hw, rel_l2(out, ref))
