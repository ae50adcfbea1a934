#!/usr/bin/env python3
"""Uninitialised-read hunt: fill the caching allocator's free blocks with 0xFF bytes (fp16 / bf16 NaN, e5m2 NaN, fp32
NaN) before running the grouped step; any buffer that is consumed without having been written turns the outputs into
NaN.  Usage: python tools/poison_check.py [--latent 64] [--batch 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.fused import GroupedDualStreamStep  # noqa: E402


def poison(gb, dev, byte=0xFF):
    blocks = []
    # many sizes, so that every size class of the caching allocator holds poisoned blocks
    for mb in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024] * 3:
        try:
            blocks.append(torch.full((mb << 20,), byte, dtype=torch.uint8, device=dev))
        except RuntimeError:
            break
    big = []
    for _ in range(int(gb)):
        big.append(torch.full((1 << 30,), byte, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    del blocks, big


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--gb", type=int, default=24)
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
    with torch.no_grad():
        ref = step(x, c, ehs, ti, ta)
        ref = {k: v.float().clone() for k, v in ref.items()}
        for byte in (0xFF, 0x7B, 0x3C):
            poison(a.gb, dev, byte)
            out = step(x, c, ehs, ti, ta)
            for k in ref:
                o = out[k].float()
                nan = int(torch.isnan(o).sum())
                err = float((o - ref[k]).norm() / ref[k].norm()) if nan == 0 else float("nan")
                print(f"poison 0x{byte:02X}: {k}: NaNs {nan}, rel-L2 vs first run {err:.3e}", flush=True)


if __name__ == "__main__":
    main()
