#!/usr/bin/env python3
"""Which parameters carry the fp16-checkpoint error?  ("keep the few most sensitive weights --
conv_out, the 26 exchange 1x1s, the time-embedding MLP -- as (hi, lo) pairs; measure and state the result either way".)

Pure CPU experiment on the ORACLE (fp32 arithmetic): round every parameter to fp16 EXCEPT a chosen subset kept in fp32 --
what a (hi, lo) pair for that subset would give the product at best -- and compare the dual-stream step with the
all-fp32 oracle.  SD-1.x-size networks, one 16x16 latent (the figure barely depends on the latent size, DESIGN.md
section 5).

    python tools/weight_sensitivity.py [--latent 16] [--out profiles/r04_weight_sensitivity.json]
"""
import argparse
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unirenderer_oracle as O  # noqa: E402  (a measurement tool, not the product)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_weight_sensitivity.json"))
    a = ap.parse_args()
    oracle = O.build_triplet(O.SD15_CONFIG, seed=1234)
    x, c, ehs, ti, ta = O.make_inputs(1, a.latent, 768, seed=99)
    ref = O.dual_stream_step(*oracle, x, c, ehs, ti, ta)
    subsets = {
        "nothing kept (all parameters fp16-rounded)": lambda n: False,
        "conv_out + conv_norm_out": lambda n: n.startswith(("conv_out", "conv_norm_out")),
        "+ the 26 exchange 1x1 convs": lambda n: n.startswith(("conv_out", "conv_norm_out", "controlnet_down_blocks", "controlnet_mid_block",
                                                                "control_down_blocks", "control_mid_block")),
        "+ time-embedding MLP and every time_emb_proj": lambda n: n.startswith(("conv_out", "conv_norm_out", "controlnet_down_blocks",
                                                                                 "controlnet_mid_block", "control_down_blocks",
                                                                                 "control_mid_block", "time_embedding")) or "time_emb_proj" in n,
        "+ conv_in and every norm / bias vector": lambda n: n.startswith(("conv_out", "conv_norm_out", "controlnet_down_blocks",
                                                                         "controlnet_mid_block", "control_down_blocks", "control_mid_block",
                                                                         "time_embedding", "conv_in")) or "time_emb_proj" in n or "norm" in n
                                                            or n.endswith(".bias"),
        "all 3x3 conv weights kept (resnets, samplers)": lambda n: n.endswith("weight") and (".conv1." in n or ".conv2." in n or ".conv." in n),
        "all attention / feed-forward weights kept": lambda n: ".attentions." in n,
    }
    rows = []
    for name, keep in subsets.items():
        q = copy.deepcopy(oracle)
        kept = total = 0
        for m in q:
            for n, p in m.named_parameters():
                total += p.numel()
                if keep(n):
                    kept += p.numel()
                else:
                    p.data = p.data.to(torch.float16).to(torch.float32)
        out = O.dual_stream_step(*q, x, c, ehs, ti, ta)
        row = dict(kept_in_fp32=name, kept_fraction_of_parameters=round(kept / total, 4),
                   rel_l2_img_pred=rel(out["img_pred"], ref["img_pred"]), rel_l2_attr_pred=rel(out["attr_pred"], ref["attr_pred"]))
        rows.append(row)
        print(json.dumps(row), flush=True)
        del q
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(what="fp16 rounding of the checkpoint, by parameter subset kept in fp32 (oracle arithmetic fp32); SD-1.x size, "
                        f"random init, batch 1, {a.latent}x{a.latent} latent", rows=rows), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
