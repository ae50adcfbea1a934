#!/usr/bin/env python3
"""Run the grouped step N times on the same inputs and count runs whose outputs are not bit-identical to the first."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.fused import GroupedDualStreamStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--churn", type=int, default=1, help="allocate / free odd-sized blocks between runs (moves tensors around)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    unet, enc, dec = bench.build_models(dev, torch.float16)
    B, L = a.batch, a.latent
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, 4, L, L, device=dev, generator=g).half()
    c = torch.randn(B, 28, L, L, device=dev, generator=g).half()
    ehs = (torch.randn(B, 77, 768, device=dev, generator=g) * 0.5).half()
    ti = torch.randint(0, 1000, (B,), device=dev, generator=g)
    ta = torch.randint(0, 1000, (B,), device=dev, generator=g)
    step = GroupedDualStreamStep(unet, enc, dec)
    bad = 0
    worst = 0.0
    with torch.no_grad():
        ref = {k: v.clone() for k, v in step(x, c, ehs, ti, ta).items()}
        keep = []
        for i in range(a.n):
            if a.churn:
                keep.append(torch.full(((i * 7919) % 50 + 1, 1 << 20), 0x7B, dtype=torch.uint8, device=dev))
                if len(keep) > 5:
                    keep.pop(0)
            out = step(x, c, ehs, ti, ta)
            same = all(torch.equal(out[k], ref[k]) for k in ref)
            if not same:
                bad += 1
                e = max(float((out[k].float() - ref[k].float()).norm() / ref[k].float().norm()) for k in ref)
                worst = max(worst, e)
                print(f"run {i}: differs, rel-L2 {e:.3e}", flush=True)
    print(f"{bad} of {a.n} runs differ from the first (worst {worst:.3e})")


if __name__ == "__main__":
    main()
