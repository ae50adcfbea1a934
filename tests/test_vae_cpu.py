"""CPU tests pinning the VAE restatement (oracle/vae_oracle.py) and the product's parameter surface
(uni_renderer_amd/vae.py): known parameter count of the SD-1.x AutoencoderKL, diffusers key names, op-level identities
of the two non-obvious leaves (single-head attention, asymmetric-pad downsample), posterior object semantics."""
import torch
import torch.nn.functional as F

from util_models import ROOT  # noqa: F401  (path setup)

from oracle import vae_oracle as V


def test_sd_vae_parameter_count_and_keys():
    from uni_renderer_amd.vae import AutoencoderKL

    o = V.build(V.SD_VAE_CONFIG)
    assert sum(p.numel() for p in o.parameters()) == 83_653_863  # the SD-1.x AutoencoderKL
    m = AutoencoderKL()
    assert sum(p.numel() for p in m.parameters()) == 83_653_863
    so, sm = o.state_dict(), m.state_dict()
    assert so.keys() == sm.keys() and all(so[k].shape == sm[k].shape for k in so)
    for k in ("encoder.down_blocks.0.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_q.bias",
              "encoder.mid_block.attentions.0.group_norm.weight", "decoder.up_blocks.0.upsamplers.0.conv.weight",
              "decoder.up_blocks.3.resnets.2.conv2.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",
              "quant_conv.weight", "post_quant_conv.bias", "encoder.conv_norm_out.weight", "decoder.conv_out.bias"):
        assert k in sm, k
    assert m.config["scaling_factor"] == 0.18215 and m.config["latent_channels"] == 4
    assert not any(k.startswith("decoder.up_blocks.3.upsamplers") or k.startswith("encoder.down_blocks.3.downsamplers") for k in sm)


def test_oracle_leaves_against_torch_functional():
    torch.manual_seed(0)
    a = V.Attention(64, 32)
    x = torch.randn(2, 64, 6, 5)
    t = a.group_norm(x).view(2, 64, 30).transpose(1, 2)
    q, k, v = a.to_q(t), a.to_k(t), a.to_v(t)
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]  # one head of dim C
    ref = a.to_out[0](ref).transpose(1, 2).reshape(2, 64, 6, 5) + x
    assert torch.allclose(a(x), ref, atol=1e-5)
    d = V.Downsample2D(64)
    y = d(x)
    xp = torch.zeros(2, 64, 7, 6)
    xp[:, :, :6, :5] = x  # zero row below, zero column right
    assert y.shape == (2, 64, 3, 2) and torch.allclose(y, F.conv2d(xp, d.conv.weight, d.conv.bias, stride=2), atol=1e-6)
    u = V.Upsample2D(64)
    assert u(x).shape == (2, 64, 12, 10)


def test_shapes_and_posterior():
    from uni_renderer_amd.vae import DiagonalGaussianDistribution

    o = V.build(V.TINY_VAE_CONFIG)
    x = torch.randn(2, 3, 32, 48)
    mean, logvar = o.encode_moments(x)
    assert mean.shape == (2, 4, 16, 24) and logvar.shape == mean.shape
    assert o.decode(mean).shape == (2, 3, 32, 48)
    p = DiagonalGaussianDistribution(torch.cat([mean, logvar * 100], 1))
    assert float(p.logvar.max()) <= 20.0 and float(p.logvar.min()) >= -30.0
    g = torch.Generator().manual_seed(1)
    s1 = p.sample(g)
    g = torch.Generator().manual_seed(1)
    assert torch.equal(s1, p.mean + p.std * torch.randn(mean.shape, generator=g)) and torch.equal(p.mode(), mean)


def test_from_pretrained_accepts_the_deprecated_attention_key_names(tmp_path):
    """SD 1.4 / 1.5 / 2.x ``vae/`` folders name the mid-block attention query / key / value / proj_attn (diffusers renames
    them on load); ADVICE r2: strict loading rejected them."""
    import json
    import os

    from safetensors.torch import save_file

    from uni_renderer_amd.vae import AutoencoderKL

    torch.manual_seed(0)
    small = dict(block_out_channels=(64, 64), down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
                 layers_per_block=1, norm_num_groups=8)
    m = AutoencoderKL(**small)
    new2old = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    sd = {}
    for k, v in m.state_dict().items():
        for new, old in new2old.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
                if k.endswith("weight") and "decoder" in k:
                    v = v[:, :, None, None]  # the still older 1x1-conv shape, on one of the two blocks
        sd[k] = v.detach().clone().contiguous()
    assert any(".query." in k for k in sd) and not any(".to_q." in k for k in sd)
    d = tmp_path / "vae"
    os.makedirs(d)
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in m.config.items()}, open(d / "config.json", "w"))
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    m2 = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
