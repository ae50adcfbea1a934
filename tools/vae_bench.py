#!/usr/bin/env python3
"""Time the product VAE (uni_renderer_amd/vae.py) at the reference's shapes: encode of B 512x512 images (train.py
encodes 8 groups x batch 4 per step; the pipeline 2 per call) and decode of B latents (5 per inference), fp16, eager
launches timed with HIP events after a warm-up.   python tools/vae_bench.py [--batch 4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    from uni_renderer_amd.vae import AutoencoderKL

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vae = AutoencoderKL().to(dev).half().eval()
    x = torch.randn(args.batch, 3, args.size, args.size, device=dev).half()
    z = torch.randn(args.batch, 4, args.size // 8, args.size // 8, device=dev).half()
    out = {}
    for name, fn in (("encode", lambda: vae.encode(x).latent_dist.mean), ("decode", lambda: vae.decode(z, return_dict=False)[0])):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_ms"] = round(e0.elapsed_time(e1) / 5, 3)
    # algorithmic FLOP of the SD-1.x VAE at 512x512 (2*MAC): encoder ~1.24 TFLOP, decoder ~2.54 TFLOP per image
    out.update(batch=args.batch, size=args.size, peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
