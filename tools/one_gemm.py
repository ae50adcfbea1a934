#!/usr/bin/env python3
"""Run ONE implicit-GEMM problem repeatedly (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/one_gemm.py M N K [--taps 1|9] [--tile T] [--splitk S] [--iters 50] [--hw H] [--batch B]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--taps", type=int, default=1)
    ap.add_argument("--tile", type=int, default=None)
    ap.add_argument("--splitk", type=int, default=None)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="grouped launch of S problems (the product's z = 2)")
    args = ap.parse_args()
    from uni_renderer_amd import ops

    dev = torch.device("cuda:0")
    dt = torch.float16
    if args.taps == 1:
        x = torch.randn(args.M, args.K, device=dev).to(dt)
        w = (torch.randn(args.N, args.K, device=dev) * 0.05).to(dt)
        r = torch.randn(args.M, args.N, device=dev).to(dt) if args.res else None
        b = torch.randn(args.N, device=dev)
        fn = lambda: ops.linear(x, w, b, res=r, tile=args.tile, splitk=args.splitk)
    else:
        cin = args.K // 9
        B, S = 4, args.streams
        hw = int(round((args.M // B) ** 0.5))
        x = torch.randn(S * B, hw, hw, cin, device=dev).to(dt)
        w = (torch.randn(S, args.N, args.K, device=dev) * 0.02).to(dt) if S > 1 else (torch.randn(args.N, args.K, device=dev) * 0.02).to(dt)
        b = torch.randn(S, args.N, device=dev) if S > 1 else torch.randn(args.N, device=dev)
        fn = lambda: ops.conv3x3(x, w, b, tile=args.tile, splitk=args.splitk, streams=S)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    print(f"M={args.M} N={args.N} K={args.K} taps={args.taps} tile={args.tile} splitk={args.splitk}: {us:.2f} us/launch "
          f"(host-paced), {2.0 * args.M * args.N * args.K / us / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
