#!/usr/bin/env python3
"""ur_wgrad against the transposed-operand path on the weight-gradient problems of the cfg 4 training step (B = 4, 64x64
latent, SD-size UNet): microseconds per call and useful TFLOP/s, per tile / slice count.

    python tools/wgrad_bench.py [--iters 20] [--sweep]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import backward as B_  # noqa: E402

CONV = [(4, 64, 320, 320, 1), (4, 64, 640, 320, 1), (4, 64, 960, 320, 1), (4, 64, 320, 320, 2), (4, 32, 320, 640, 1), (4, 32, 640, 640, 1),
        (4, 32, 1280, 640, 1), (4, 32, 960, 640, 1), (4, 32, 1920, 640, 1), (4, 16, 640, 1280, 1), (4, 16, 1280, 1280, 1), (4, 16, 2560, 1280, 1),
        (4, 16, 1920, 1280, 1), (4, 8, 1280, 1280, 1), (4, 8, 2560, 1280, 1)]
LIN = [(16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280),
       (1024, 10240, 1280), (1024, 1280, 5120), (308, 320, 768), (308, 640, 768), (308, 1280, 768), (16384, 960, 320), (4, 1280, 320)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sweep", action="store_true", help="also time every tile and a few slice counts")
    ap.add_argument("--tiles", default="1,4,5,6")
    ap.add_argument("--splits", default="0,1,2,4,8,16")
    args = ap.parse_args()
    dt = torch.bfloat16
    TILES, SPLITS = [int(v) for v in args.tiles.split(",")], [int(v) for v in args.splits.split(",")]
    mk = lambda *s: torch.randn(*s, device="cuda").to(dt)
    tot = {"wgrad": 0.0, "old": 0.0}
    for (Bn, H, Cc, N, stride) in CONV:
        Ho = H // stride
        x, dy, w = mk(Bn, H, H, Cc), mk(Bn, Ho, Ho, N), mk(N, 9 * Cc)
        fl = 2.0 * Bn * Ho * Ho * N * 9 * Cc
        row = f"conv B{Bn} {H}x{H} C{Cc} N{N} s{stride}"
        t_new = timeit(lambda: B_.wgrad(dy.reshape(-1, N), x, True, conv=(Ho, Ho, stride)), args.iters)

        def old():
            B_.WGRAD = False
            try:
                B_.conv3x3_backward(x, w, dy, True, stride, need_dx=False)
            finally:
                B_.WGRAD = True
        t_old = timeit(old, args.iters)
        tot["wgrad"] += t_new
        tot["old"] += t_old
        extra = ""
        if args.sweep:
            for tile in TILES:
                for sp in SPLITS:
                    t = timeit(lambda: B_.wgrad(dy.reshape(-1, N), x, True, conv=(Ho, Ho, stride), tile=tile, splits=sp), args.iters)
                    extra += f" t{tile}s{sp}:{t:.0f}"
        print(f"{row:34s} wgrad {t_new:7.1f} us {fl / t_new / 1e6:6.1f} TF | transposed path {t_old:7.1f} us{extra}", flush=True)
    for (P, N, K) in LIN:
        x, dy = mk(P, K), mk(P, N)
        fl = 2.0 * P * N * K
        t_new = timeit(lambda: B_.wgrad(dy, x, True), args.iters)

        def old():
            (dyt, xt), db = B_.transpose2d_many([dy, x], colsum_of=0, pad64=(0, 1))
            B_.ops.linear(dyt, xt)
        t_old = timeit(old, args.iters)
        tot["wgrad"] += t_new
        tot["old"] += t_old
        extra = ""
        if args.sweep:
            for tile in TILES:
                for sp in SPLITS:
                    t = timeit(lambda: B_.wgrad(dy, x, True, tile=tile, splits=sp), args.iters)
                    extra += f" t{tile}s{sp}:{t:.0f}"
        print(f"linear P{P} N{N} K{K:24d} wgrad {t_new:7.1f} us {fl / t_new / 1e6:6.1f} TF | transposed path {t_old:7.1f} us{extra}", flush=True)
    print(f"sum: wgrad {tot['wgrad']:.0f} us, transposed path {tot['old']:.0f} us")


if __name__ == "__main__":
    main()
