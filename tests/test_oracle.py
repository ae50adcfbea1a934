"""CPU tests that PIN the oracle (oracle/unirenderer_oracle.py).  The reference has no tests or golden
tensors for this path and cannot be imported, so the oracle is pinned by: known-answer
parameter counts, the reference's structural invariants, op-level cross-checks of every leaf against
torch.nn.functional, and the committed golden vectors."""
import json
import math
import os

import torch
import torch.nn.functional as F

from util_models import O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answer_parameter_counts():
    # SURVEY.md §7 step 1 / §8a a1,a3,a5: SD-1.x unet; enc/dec before and after the 4->28 channel surgery
    with torch.device("meta"):
        u = O.UNet2DConditionModel(**O.SD15_CONFIG)
        e4, d4 = O.AttributeEncoderModel(**O.SD15_CONFIG), O.AttributeDecoderModel(**O.SD15_CONFIG)
        e28 = O.AttributeEncoderModel(**dict(O.SD15_CONFIG, in_channels=28))
        d28 = O.AttributeDecoderModel(**dict(O.SD15_CONFIG, out_channels=28))
    assert O.count_params(u) == 859_520_964
    assert O.count_params(e4) == 360_192_640 and O.count_params(d4) == 524_338_244
    assert O.count_params(e28) == 360_261_760 and O.count_params(d28) == 524_407_388


def test_arity_and_skip_shapes_tiny():
    unet, enc, dec = O.build_triplet(O.TINY_CONFIG, seed=1)
    x, c, ehs, ti, ta = O.make_inputs(2, 16, 64)
    out = O.dual_stream_step(unet, enc, dec, x, c, ehs, ti, ta)
    assert len(out["raw_unet"]) == 12 and len(out["raw_enc"]) == 12 and len(out["enc_res"]) == 12
    assert len(out["up_res"]) == 13
    shapes = [tuple(t.shape[1:]) for t in out["raw_unet"]]
    assert shapes == [(64, 16, 16)] * 3 + [(64, 8, 8)] + [(128, 8, 8)] * 2 + [(128, 4, 4)] + [(128, 4, 4)] * 2 + \
        [(128, 2, 2)] * 3
    ups = [tuple(t.shape[1:]) for t in out["up_res"]]
    assert ups == [(128, 2, 2)] * 4 + [(128, 4, 4)] * 3 + [(128, 8, 8)] * 3 + [(64, 16, 16)] * 3
    assert out["img_pred"].shape == (2, 4, 16, 16) and out["attr_pred"].shape == (2, 28, 16, 16)


def test_reference_invariants():
    unet, enc, dec = O.build_triplet(O.TINY_CONFIG, seed=2, exchange_std=0.0)
    x, c, ehs, ti, ta = O.make_inputs(1, 16, 64)
    with torch.no_grad():
        r1 = enc(x, ta, ehs, controlnet_cond=c)
        r2 = enc(torch.randn_like(x), ta, ehs, controlnet_cond=c)  # enc ignores `sample` (controlnet.py:1716-1720)
        assert torch.equal(r1[3], r2[3])
        assert all(float(t.abs().max()) == 0 for t in r1[0])  # zero-init exchange convs
        a = unet(x, ti, ehs, r1[0], r1[1])
        b = unet(x, ti, ehs)
        assert torch.equal(a[0], b[0])
        d1 = dec(r1[3], r1[2], ta, ehs, a[1], a[2])
        d2 = dec(r1[3], r1[2], ta, ehs, tuple(torch.zeros_like(t) for t in a[1]), torch.zeros_like(a[2]))
        assert torch.equal(d1, d2)


def test_exchange_is_live_when_randomised():
    unet, enc, dec = O.build_triplet(O.TINY_CONFIG, seed=2, exchange_std=0.02)
    x, c, ehs, ti, ta = O.make_inputs(1, 16, 64)
    with torch.no_grad():
        r = enc(x, ta, ehs, controlnet_cond=c)
        a = unet(x, ti, ehs, r[0], r[1])
        b = unet(x, ti, ehs)
    assert float((a[0] - b[0]).abs().max()) > 1e-4


def test_timestep_known_answers():
    e = O.timestep_sinusoid(torch.tensor([0]), 320, True, 0)
    assert torch.equal(e[0], torch.cat([torch.ones(160), torch.zeros(160)]))  # Timesteps(t=0) = [1]*160 + [0]*160
    e = O.timestep_sinusoid(torch.tensor([7]), 8, False, 0)
    f = torch.exp(-math.log(10000.0) * torch.arange(4) / 4)
    assert torch.allclose(e[0], torch.cat([torch.sin(7 * f), torch.cos(7 * f)]))


def test_leaves_against_functional():
    torch.manual_seed(0)
    # Attention == F.scaled_dot_product_attention
    a = O.Attention(64, 2, 32).eval()
    x = torch.randn(2, 50, 64)
    q, k, v = [t.view(2, 50, 2, 32).transpose(1, 2) for t in (a.to_q(x), a.to_k(x), a.to_v(x))]
    ref = a.to_out[0](F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 50, 64))
    assert torch.allclose(a(x), ref, atol=1e-5)
    # cross attention with 77 keys of another width
    a2 = O.Attention(64, 2, 32, cross_attention_dim=48).eval()
    ctx = torch.randn(2, 77, 48)
    q, k, v = a2.to_q(x), a2.to_k(ctx), a2.to_v(ctx)
    q, k, v = [t.view(2, -1, 2, 32).transpose(1, 2) for t in (q, k, v)]
    ref = a2.to_out[0](F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 50, 64))
    assert torch.allclose(a2(x, ctx), ref, atol=1e-5)
    # GEGLU: erf gelu on the SECOND half
    g = O.GEGLU(16, 32).eval()
    y = g.proj(torch.ones(1, 16))
    assert torch.allclose(g(torch.ones(1, 16)), y[:, :32] * 0.5 * y[:, 32:] * (1 + torch.erf(y[:, 32:] / math.sqrt(2))), atol=1e-6)
    # Resnet: explicit formula
    r = O.ResnetBlock2D(64, 128, 32).eval()
    xr, tr = torch.randn(1, 64, 4, 4), torch.randn(1, 32)
    h = F.conv2d(F.silu(F.group_norm(xr, 32, r.norm1.weight, r.norm1.bias, 1e-5)), r.conv1.weight, r.conv1.bias, padding=1)
    h = h + F.linear(F.silu(tr), r.time_emb_proj.weight, r.time_emb_proj.bias)[:, :, None, None]
    h = F.conv2d(F.silu(F.group_norm(h, 32, r.norm2.weight, r.norm2.bias, 1e-5)), r.conv2.weight, r.conv2.bias, padding=1)
    ref = F.conv2d(xr, r.conv_shortcut.weight, r.conv_shortcut.bias) + h
    assert torch.allclose(r(xr, tr), ref, atol=1e-5)
    # Up / down sample
    u = O.Upsample2D(8).eval()
    xu = torch.randn(1, 8, 3, 5)
    assert torch.allclose(u(xu), F.conv2d(xu.repeat_interleave(2, 2).repeat_interleave(2, 3), u.conv.weight, u.conv.bias, padding=1), atol=1e-6)
    d = O.Downsample2D(8).eval()
    assert d(torch.randn(1, 8, 6, 6)).shape == (1, 8, 3, 3)


def test_diffusers_state_dict_key_names():
    with torch.device("meta"):
        u = O.UNet2DConditionModel(**O.SD15_CONFIG)
    keys = set(u.state_dict().keys())
    for k in [
        "conv_in.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias",
        "down_blocks.0.resnets.0.norm1.weight", "down_blocks.0.resnets.1.time_emb_proj.bias",
        "down_blocks.0.attentions.0.proj_in.weight", "down_blocks.0.attentions.1.norm.bias",
        "down_blocks.1.resnets.0.conv_shortcut.weight", "down_blocks.0.downsamplers.0.conv.weight",
        "down_blocks.2.attentions.0.transformer_blocks.0.attn1.to_q.weight",
        "down_blocks.2.attentions.0.transformer_blocks.0.attn2.to_k.weight",
        "down_blocks.2.attentions.0.transformer_blocks.0.attn1.to_out.0.bias",
        "down_blocks.2.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
        "down_blocks.2.attentions.0.transformer_blocks.0.ff.net.2.bias",
        "down_blocks.2.attentions.0.transformer_blocks.0.norm3.weight",
        "mid_block.attentions.0.proj_out.weight", "mid_block.resnets.1.conv2.weight",
        "up_blocks.0.resnets.2.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight",
        "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_v.weight", "conv_norm_out.weight", "conv_out.bias",
    ]:
        assert k in keys, k
    assert u.state_dict()["down_blocks.0.attentions.0.proj_in.weight"].shape == (320, 320, 1, 1)
    assert u.state_dict()["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (320, 768)
    assert u.state_dict()["up_blocks.0.resnets.0.conv1.weight"].shape == (1280, 2560, 3, 3)
    with torch.device("meta"):
        e, d = O.AttributeEncoderModel(**O.SD15_CONFIG), O.AttributeDecoderModel(**O.SD15_CONFIG)
    assert {"controlnet_down_blocks.11.weight", "controlnet_mid_block.bias"} <= set(e.state_dict())
    assert {"control_down_blocks.0.weight", "control_mid_block.weight"} <= set(d.state_dict())
    assert not any(k.startswith("up_blocks") for k in e.state_dict())
    assert not any(k.startswith(("down_blocks", "conv_in", "mid_block")) for k in d.state_dict())


def test_committed_golden_vectors():
    from safetensors.torch import load_file

    g = load_file(os.path.join(GOLD, "tiny_step.safetensors"))
    unet, enc, dec = O.build_triplet(O.TINY_CONFIG, seed=1234)
    chk = json.load(open(os.path.join(GOLD, "tiny_weights_checksum.json")))
    for name, m in (("unet", unet), ("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            s, a = chk[f"{name}.{k}"]
            assert abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), f"seeded init drifted: {name}.{k}"
    out = O.dual_stream_step(unet, enc, dec, g["x_t"], g["cond"], g["ehs"], g["t_img"], g["t_attr"])
    for k in ("img_pred", "attr_pred", "raw_mid_unet", "raw_mid_enc", "enc_mid"):
        assert torch.allclose(out[k], g[k], rtol=1e-4, atol=1e-5), k
    assert torch.allclose(out["enc_res"][11], g["enc_res_11"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(out["up_res"][12], g["up_res_12"], rtol=1e-4, atol=1e-5)


def test_committed_leaf_vectors():
    from safetensors.torch import load_file

    g = load_file(os.path.join(GOLD, "leaf_vectors.safetensors"))

    def load(m, prefix):
        m.load_state_dict({k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)})
        return m.eval()

    with torch.no_grad():
        r = load(O.ResnetBlock2D(128, 64, 256), "w.resnet.")
        assert torch.allclose(r(g["resnet_x"], g["resnet_temb"]), g["resnet_y"], rtol=1e-4, atol=1e-5)
        t = load(O.Transformer2DModel(2, 32, 64, 64), "w.tfm.")
        assert torch.allclose(t(g["tfm_x"], g["tfm_ctx"]), g["tfm_y"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(load(O.Upsample2D(64), "w.up.")(g["tfm_x"]), g["up_y"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(load(O.Downsample2D(64), "w.down.")(g["tfm_x"]), g["down_y"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(O.timestep_sinusoid(torch.tensor([0, 1, 500, 999]), 320, True, 0), g["timestep_emb"], atol=1e-6)
