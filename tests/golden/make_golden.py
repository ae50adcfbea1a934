#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (oracle/unirenderer_oracle.py).

    python tests/golden/make_golden.py

The reference itself cannot be imported here (diffusers 0.24 absent), so these vectors pin
the ORACLE (and through it the GPU path), not the reference: "parity unpinned" stays in force.

Outputs:
  tiny_step.safetensors      inputs + outputs of one dual-stream step, TINY_CONFIG, B=2, 16x16 latent
  tiny_weights_checksum.json per-tensor (sum, abs-sum) of the seeded weights, to detect RNG / init drift
  leaf_vectors.safetensors   inputs/outputs of single leaves (resnet, transformer, up/down sample, timestep)
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import unirenderer_oracle as O  # noqa: E402


def main():
    from safetensors.torch import save_file

    torch.set_num_threads(1)  # deterministic summation order
    unet, enc, dec = O.build_triplet(O.TINY_CONFIG, seed=1234)
    x, c, ehs, ti, ta = O.make_inputs(2, 16, 64, seed=99)
    out = O.dual_stream_step(unet, enc, dec, x, c, ehs, ti, ta)
    blob = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, img_pred=out["img_pred"], attr_pred=out["attr_pred"],
                raw_mid_unet=out["raw_mid_unet"], raw_mid_enc=out["raw_mid_enc"], enc_mid=out["enc_mid"],
                enc_res_0=out["enc_res"][0], enc_res_11=out["enc_res"][11], raw_unet_5=out["raw_unet"][5],
                up_res_12=out["up_res"][12])
    save_file({k: v.contiguous() for k, v in blob.items()}, os.path.join(HERE, "tiny_step.safetensors"))
    chk = {}
    for name, m in (("unet", unet), ("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            chk[f"{name}.{k}"] = [float(v.double().sum()), float(v.double().abs().sum())]
    with open(os.path.join(HERE, "tiny_weights_checksum.json"), "w") as f:
        json.dump(chk, f)

    torch.manual_seed(77)
    g = torch.Generator().manual_seed(78)
    leaf = {}
    r = O.ResnetBlock2D(128, 64, 256).eval()
    xr, tr = torch.randn(2, 128, 8, 8, generator=g), torch.randn(2, 256, generator=g)
    t2 = O.Transformer2DModel(2, 32, 64, 64).eval()
    xt, ct = torch.randn(2, 64, 8, 8, generator=g), torch.randn(2, 77, 64, generator=g)
    up, down = O.Upsample2D(64).eval(), O.Downsample2D(64).eval()
    with torch.no_grad():
        leaf.update(resnet_x=xr, resnet_temb=tr, resnet_y=r(xr, tr))
        leaf.update(tfm_x=xt, tfm_ctx=ct, tfm_y=t2(xt, ct))
        leaf.update(up_y=up(xt), down_y=down(xt))
        leaf["timestep_emb"] = O.timestep_sinusoid(torch.tensor([0, 1, 500, 999]), 320, True, 0)
    for name, m in (("resnet", r), ("tfm", t2), ("up", up), ("down", down)):
        for k, v in m.state_dict().items():
            leaf[f"w.{name}.{k}"] = v
    save_file({k: v.contiguous() for k, v in leaf.items()}, os.path.join(HERE, "leaf_vectors.safetensors"))
    print("wrote golden vectors to", HERE)


if __name__ == "__main__":
    main()
