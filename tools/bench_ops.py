#!/usr/bin/env python3
"""Micro-benchmarks of the non-GEMM device ops at the shapes of the headline step (B=4, 64x64 latent), timed as
HIP-graph replays of 20 launches (median of 5 replays).  Used to choose launch heuristics.

    python tools/bench_ops.py [--what gn,ln,attn] > gpurun_out/bench_ops.jsonl
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, rounds=5, iters=20):
    """median us per launch over `rounds` replays of a HIP graph holding `iters` launches (no host pacing)."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gn,ln,attn")
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    from uni_renderer_amd import ops

    dev = torch.device("cuda:0")
    dt = torch.float16
    B = args.batch
    what = args.what.split(",")
    if "gn" in what:
        shapes = [(320, 0, 4096), (640, 320, 4096), (320, 0, 1024), (640, 0, 1024), (1280, 640, 1024), (1280, 0, 256),
                  (1280, 1280, 256), (1280, 0, 64), (1280, 1280, 64)]
        for c0, c1, rows in shapes:
            x0 = torch.randn(B, rows, 1, c0, device=dev).to(dt)
            x1 = torch.randn(B, rows, 1, c1, device=dev).to(dt) if c1 else None
            g = torch.ones(c0 + c1, device=dev)
            b = torch.zeros(c0 + c1, device=dev)
            nbytes = B * rows * (c0 + c1) * 2
            best = None
            for nstat in (None, 8, 16, 32, 64, 128):
                for napply in (None, 32, 64, 128, 256, 512):
                    if (nstat or 1) > rows // 8 or (napply or 1) > rows:
                        continue
                    us = timeit(lambda: ops.groupnorm(x0, g, b, 1e-5, x1=x1, silu=True, nstat=nstat, napply=napply))
                    rec = dict(op="groupnorm", c0=c0, c1=c1, rows=rows, nstat=nstat, napply=napply, us=round(us, 2),
                               gbs=round(3 * nbytes / us / 1e3, 1))
                    print(json.dumps(rec), flush=True)
                    if best is None or us < best["us"]:
                        best = rec
            print(json.dumps(dict(best, op="groupnorm_best")), flush=True)
    if "ln" in what:
        for C, rows in [(320, 4096 * B), (640, 1024 * B), (1280, 256 * B), (1280, 64 * B)]:
            x = torch.randn(rows, C, device=dev).to(dt)
            g = torch.ones(C, device=dev)
            b = torch.zeros(C, device=dev)
            us = timeit(lambda: ops.layernorm(x, g, b))
            print(json.dumps(dict(op="layernorm", C=C, rows=rows, us=round(us, 2), gbs=round(2 * rows * C * 2 / us / 1e3, 1))),
                  flush=True)
    if "attn" in what:
        for T, Tk, d in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40),
                         (1024, 77, 80), (256, 77, 160)]:
            H = 8
            C = H * d
            q = torch.randn(B, T, C, device=dev).to(dt)
            k = torch.randn(B, Tk, C, device=dev).to(dt)
            Tp = (Tk + 63) // 64 * 64
            vt = torch.zeros(B, C, Tp, device=dev, dtype=dt)
            vt[:, :, :Tk] = torch.randn(B, C, Tk, device=dev).to(dt)
            us = timeit(lambda: ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C))
            fl = 4.0 * B * H * T * Tk * d
            print(json.dumps(dict(op="attention", T=T, Tk=Tk, d=d, us=round(us, 2), tflops=round(fl / us / 1e6, 1))), flush=True)
    if "add" in what:
        for n in (B * 4096 * 320, B * 64 * 1280):
            a = torch.randn(n, device=dev).to(dt)
            b2 = torch.randn(n, device=dev).to(dt)
            us = timeit(lambda: ops.add(a, b2))
            print(json.dumps(dict(op="add", n=n, us=round(us, 2), gbs=round(3 * n * 2 / us / 1e3, 1))), flush=True)


if __name__ == "__main__":
    main()
