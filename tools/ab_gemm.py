#!/usr/bin/env python3
"""Kernel-level A/B of ur_igemm at the heaviest problems of the headline step (z = 2 grouped launches).

Each problem is captured into a HIP graph of 20 launches and replayed (no host pacing); the reported time is the
median over 5 replays / 20.  Run once per library build for a same-box comparison:

    python tools/ab_gemm.py                      # uni_renderer_amd/liburhip.so
    UR_LIB_PATH=/path/to/other.so python tools/ab_gemm.py
    python tools/ab_gemm.py --sweep              # every tile x split-K for each problem (slow)
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import ops  # noqa: E402

# (M per stream, N, K, taps, streams); N >= 2560 are the feed-forward input GEMMs (GEGLU epilogue, no residual)
PROBLEMS = [
    (16384, 320, 320, 1, 2), (16384, 320, 2880, 9, 2), (16384, 320, 5760, 9, 2), (16384, 2560, 320, 1, 2),
    (16384, 320, 1280, 1, 2), (16384, 640, 320, 1, 2),
    (4096, 640, 640, 1, 2), (4096, 640, 5760, 9, 2), (4096, 5120, 640, 1, 2), (4096, 640, 2560, 1, 2),
    (4096, 1280, 640, 1, 2),
    (1024, 1280, 1280, 1, 2), (1024, 1280, 11520, 9, 2), (1024, 10240, 1280, 1, 2), (1024, 1280, 5120, 1, 2),
    (256, 1280, 11520, 9, 2), (256, 1280, 1280, 1, 2), (256, 1280, 23040, 9, 2),
]


def build(M, N, K, taps, S, dev, dt):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    if taps == 1:
        x = torch.randn(S * M, K, generator=g).to(dev).to(dt)
        w = (torch.randn(S, N, K, generator=g) * K ** -0.5).to(dev).to(dt)
        b = torch.randn(S, N, generator=g).to(dev)
        if N >= 2560:
            return lambda tile, sk: ops.linear(x, w, b, act=ops.ACT_GEGLU, tile=tile, splitk=sk, streams=S)
        r = torch.randn(S * M, N, generator=g).to(dev).to(dt)
        return lambda tile, sk: ops.linear(x, w, b, res=r, tile=tile, splitk=sk, streams=S)
    cin = K // 9
    B = 4
    hw = int(round((M // B) ** 0.5))
    assert B * hw * hw == M
    x = torch.randn(S * B, hw, hw, cin, generator=g).to(dev).to(dt)
    w = (torch.randn(S, N, K, generator=g) * K ** -0.5).to(dev).to(dt)
    b = torch.randn(S, N, generator=g).to(dev)
    return lambda tile, sk: ops.conv3x3(x, w, b, tile=tile, splitk=sk, streams=S)


def time_graph(fn, reps=20, rounds=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
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
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tiles", default="", help="comma-separated tile ids for --sweep (default: all)")
    ap.add_argument("--problems", default="", help='subset, e.g. "256,1280,11520,9,2;1024,1280,11520,9,2"')
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    total = 0.0
    rows = []
    problems = [tuple(int(v) for v in q.split(",")) for q in args.problems.split(";")] if args.problems else PROBLEMS
    for (M, N, K, taps, S) in problems:
        fn = build(M, N, K, taps, S, dev, dt)
        fl = 2.0 * M * N * K * S
        if args.sweep:
            res = {}
            for tile in ([int(t) for t in args.tiles.split(",")] if args.tiles else ops._TILES):
                for sk in (1, 2, 4, 8, 16):
                    if sk > 1 and K // 64 < 4 * sk:
                        continue
                    try:
                        res[(tile, sk)] = time_graph(lambda: fn(tile, sk), reps=10, rounds=3)
                    except RuntimeError:
                        pass
            best = min(res, key=res.get)
            us, cfg = res[best], list(best)
        else:
            cfg = list(ops.plan_igemm(M, N, K, taps, S))
            us = time_graph(lambda: fn(None, None))
        total += us
        rows.append(dict(M=M, N=N, K=K, taps=taps, z=S, cfg=cfg, us=round(us, 2), tflops=round(fl / us / 1e6, 1)))
        if args.sweep:
            rows[-1]["top"] = {f"{t},{k}": round(v, 2) for (t, k), v in sorted(res.items(), key=lambda kv: kv[1])[:6]}
        print(json.dumps(rows[-1]), flush=True)
    print(json.dumps(dict(lib=os.environ.get("UR_LIB_PATH", "default"), sum_us=round(total, 1))))
    if args.out:
        json.dump(rows, open(args.out, "w"))


if __name__ == "__main__":
    main()
