"""GPU parity of the VAE (uni_renderer_amd/vae.py) against the CPU fp32 oracle (oracle/vae_oracle.py)
on the same weights: posterior moments of ``encode``, images of ``decode``, the asymmetric-pad downsample op-level, the
SD-1.x-size network, and the sampling pipeline with the product VAE attached on both ends."""
import json

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from util_models import O, build_product_from_oracle

from oracle import vae_oracle as V

pytestmark = pytest.mark.gpu


def _product(oracle, dtype, dev):
    from uni_renderer_amd.vae import AutoencoderKL

    c = oracle.cfg
    m = AutoencoderKL(in_channels=c["in_channels"], out_channels=c["out_channels"], latent_channels=c["latent_channels"],
                      block_out_channels=c["block_out_channels"], layers_per_block=c["layers_per_block"],
                      norm_num_groups=c["norm_num_groups"], scaling_factor=c["scaling_factor"],
                      down_block_types=("DownEncoderBlock2D",) * len(c["block_out_channels"]),
                      up_block_types=("UpDecoderBlock2D",) * len(c["block_out_channels"]))
    m.load_state_dict(oracle.state_dict())
    return m.to(dtype).to(dev).eval()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("hw", [(32, 32), (40, 24)])
def test_conv3x3_stride2_asymmetric_pad(dev, dtype, tol, hw):
    """``ur_igemm`` pad = 0: F.pad(x, (0, 1, 0, 1)) + conv(stride 2, padding 0), even and odd output sizes."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3

    g = torch.Generator().manual_seed(3)
    H, W = hw
    x = torch.randn(2, 64, H, W, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * (64 * 9) ** -0.5
    b = torch.randn(128, generator=g)
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    ref = F.conv2d(F.pad(xq, (0, 1, 0, 1)), wq, b, stride=2)
    y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype), pack_conv3x3(w.to(dev), dtype), b.to(dev), stride=2, pad=0)
    assert tuple(y.shape) == (2, ref.shape[2], ref.shape[3], 128)
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_tiny_vae_encode_decode_vs_oracle(dev, dtype, tol):
    o = V.build(V.TINY_VAE_CONFIG, seed=11)
    m = _product(o, dtype, dev)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 32, 48, generator=g)
    mean, logvar = o.encode_moments(x)
    post = m.encode(x.to(dev).to(dtype)).latent_dist
    e_mean, e_lv = rel_l2(post.mean, mean), rel_l2(post.logvar, logvar)
    z = torch.randn(2, 4, 16, 24, generator=g)
    img = o.decode(z)
    out = m.decode(z.to(dev).to(dtype), return_dict=False)[0]
    e_dec = rel_l2(out, img)
    print(json.dumps(dict(vae="tiny", dtype=str(dtype), mean=e_mean, logvar=e_lv, decode=e_dec)))
    assert out.shape == (2, 3, 32, 48) and post.mean.shape == (2, 4, 16, 24)
    assert e_mean < tol and e_lv < tol and e_dec < tol
    # sample(): mean + std * noise with the caller's generator, then the scaling factor (train.py:1266-1270)
    gg = torch.Generator(device=dev).manual_seed(5)
    s = post.sample(gg)
    gg = torch.Generator(device=dev).manual_seed(5)
    assert torch.equal(s, post.mean + post.std * torch.randn(post.mean.shape, generator=gg, device=dev, dtype=post.mean.dtype))


def test_sd_size_vae_vs_oracle(dev):
    """The SD-1.x AutoencoderKL (83.7 M parameters, 512-channel single-head attention in both mid blocks) on a 128x128
    image (latent 16x16: T = 256 tokens) and a 256x256 decode (T = 1024), fp16."""
    o = V.build(V.SD_VAE_CONFIG, seed=13)
    m = _product(o, torch.float16, dev)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(1, 3, 128, 128, generator=g)
    mean, logvar = o.encode_moments(x)
    post = m.encode(x.to(dev).half()).latent_dist
    z = torch.randn(2, 4, 32, 32, generator=g)
    img = o.decode(z)
    out = m.decode(z.to(dev).half(), return_dict=False)[0]
    errs = dict(vae="sd", mean=rel_l2(post.mean, mean), logvar=rel_l2(post.logvar, logvar), decode=rel_l2(out, img))
    print(json.dumps(errs))
    assert out.shape == (2, 3, 256, 256)
    assert errs["mean"] < 3e-3 and errs["logvar"] < 3e-3 and errs["decode"] < 3e-3


def test_pipeline_with_the_product_vae_on_both_ends(dev):
    """models/pipeline.py:2533-2538 (encode image + mask) and 2755-2769 (decode the five image groups) through the
    product VAE, against the same flow with the oracle VAE + oracle networks."""
    from uni_renderer_amd.pipeline import ATTR_GROUPS, UniRendererPipeline
    from uni_renderer_amd.schedulers import DDIMScheduler

    unet_o, enc_o, dec_o = O.build_triplet(O.TINY_CONFIG, seed=28)
    unet, enc, dec = build_product_from_oracle(unet_o, enc_o, dec_o, torch.float16, dev)
    vo = V.build(V.TINY_VAE_CONFIG, seed=15)  # 2 levels: 32x32 image -> 16x16 latent
    vae = _product(vo, torch.float16, dev)
    pipe = UniRendererPipeline(vae=vae, unet=unet, controlnet=enc, controldec=dec)
    pipe.vae_scale_factor = 2
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(16)
    image, masks = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1, torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    ehs = torch.randn(1, 77, 64, generator=g) * 0.5
    noise = torch.randn(2, 4, 16, 16, generator=g)

    from uni_renderer_amd import vae as vae_mod

    orig_sample = vae_mod.DiagonalGaussianDistribution.sample
    vae_mod.DiagonalGaussianDistribution.sample = lambda self, generator=None: self.mean  # deterministic: the mode
    try:
        out = pipe.real_image2mask_3mod_albedo(prompt_embeds=ehs.to(dev).half(), image=image.to(dev).half(), masks=masks.to(dev).half(),
                                               latents=noise, num_inference_steps=2, guidance_scale=0.0, output_type="pt")
    finally:
        vae_mod.DiagonalGaussianDistribution.sample = orig_sample
    sf = vo.cfg["scaling_factor"]
    lat_img, lat_mask = vo.encode_moments(image)[0] * sf, vo.encode_moments(masks)[0] * sf
    s = DDIMScheduler()
    s.set_timesteps(2)
    lat = {n: noise.clone() for n in ATTR_GROUPS}
    e = ehs.repeat(2, 1, 1)
    for t in s.timesteps:
        cond = torch.cat([lat_mask] + [lat[n] for n in ATTR_GROUPS], 1)
        r = O.dual_stream_step(unet_o, enc_o, dec_o, lat_img, cond, e, torch.zeros(2).long(), t.expand(2))
        for k, n in enumerate(ATTR_GROUPS):
            lat[n] = s.step(r["attr_pred"][:, 4 + 4 * k:8 + 4 * k], t, lat[n])[0]
    assert rel_l2(out[0], lat["material"]) < 1e-2  # first return value: the material latents (ref 2808)
    for img, n in zip(out[1:], ATTR_GROUPS[1:]):
        ref = (vo.decode(lat[n] / sf) / 2 + 0.5).clamp(0, 1)
        assert img.shape == (2, 3, 32, 32) and rel_l2(img, ref) < 1.5e-2, (n, rel_l2(img, ref))
