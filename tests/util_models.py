"""Helpers shared by the model-level tests: build the oracle triplet and the product triplet with the SAME
weights (state-dict key compatibility is itself part of the contract)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import unirenderer_oracle as O  # noqa: E402  (tests may import the oracle)


def product_config(cfg: dict) -> dict:
    c = dict(cfg)
    c["attention_head_dim"] = cfg["attention_head_dim"]
    return c


def build_product_from_oracle(unet_o, enc_o, dec_o, dtype=None, device=None):
    import uni_renderer_amd as U

    cfg = product_config(unet_o.cfg)
    unet = U.UNet2DConditionModel(**cfg)
    unet.load_state_dict(unet_o.state_dict())
    enc = U.AttributeEncoderModel.from_unet(unet)
    dec = U.AttributeDecoderModel.from_unet(unet)
    # the reference's 4 -> 28 channel surgery (train/train.py:976,985,988-989,996), done the reference's way
    enc.conv_in.weight = torch.nn.Parameter(enc.conv_in.weight.repeat(1, 7, 1, 1) * 0.142)
    enc.register_to_config(in_channels=28)
    dec.conv_out.weight = torch.nn.Parameter(dec.conv_out.weight.repeat(7, 1, 1, 1) * 0.142)
    dec.conv_out.bias = torch.nn.Parameter(dec.conv_out.bias.repeat(7) * 0.142)
    dec.register_to_config(out_channels=28)
    enc.load_state_dict(enc_o.state_dict())  # brings the randomised exchange convs over
    dec.load_state_dict(dec_o.state_dict())
    mods = [unet, enc, dec]
    if dtype is not None:
        mods = [m.to(dtype) for m in mods]
    if device is not None:
        mods = [m.to(device) for m in mods]
    return [m.eval() for m in mods]


def product_step(unet, enc, dec, x_t, cond, ehs, t_img, t_attr, run_decoder=True):
    """The reference's per-step call pattern (pipeline.py:2660-2690 / train.py:1324-1354) on the product."""
    res, mid, raw_enc, raw_mid_enc = enc(x_t, t_attr, encoder_hidden_states=ehs, controlnet_cond=cond,
                                         return_dict=False)
    img_pred, raw_unet, raw_mid_unet, up_res = unet(
        x_t, t_img, encoder_hidden_states=ehs, down_block_additional_residuals=res,
        mid_block_additional_residual=mid, return_dict=False)
    attr_pred = None
    if run_decoder:
        attr_pred = dec(sample=raw_mid_enc, down_block_res_samples=raw_enc, timestep=t_attr,
                        encoder_hidden_states=ehs, down_block_additional_residuals=raw_unet,
                        mid_block_additional_residual=raw_mid_unet, return_dict=False)
    return dict(img_pred=img_pred, attr_pred=attr_pred, enc_res=res, enc_mid=mid, raw_enc=raw_enc,
                raw_mid_enc=raw_mid_enc, raw_unet=raw_unet, raw_mid_unet=raw_mid_unet, up_res=up_res)


class OracleScheduler:
    """torch-tensor adapter over oracle/schedulers_oracle.py (numpy float64, diffusers' D1s / einsum form): the host
    reference of the sampling-loop tests, so that the product's scheduler, its coefficient table and the fused HIP update
    kernels are all compared against code they share nothing with."""

    def __init__(self, kind: str, steps: int, **kw):
        from oracle import schedulers_oracle as S

        self.o = (S.UniPCOracle if kind == "unipc" else S.DDIMOracle)(**kw)
        self.o.set_timesteps(steps)
        self.timesteps = torch.from_numpy(self.o.timesteps.copy())

    def step(self, model_output, t, sample):
        out = self.o.step(model_output.detach().double().cpu().numpy(), int(t), sample.detach().double().cpu().numpy())
        return (torch.from_numpy(out).to(sample.dtype),)
