#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of ur_igemm: time one (M, N) GEMM at K = 64 .. 2560 (graph replay, z = 2) and fit
t = t0 + t1 * (K / 64).  t0 = launch + prologue + epilogue, t1 = one K chunk on the busiest CU."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ab_gemm import time_graph  # noqa: E402
from uni_renderer_amd import ops  # noqa: E402


def main():
    dev, dt, S = torch.device("cuda:0"), torch.float16, 2
    cases = [(16384, 320, 9), (4096, 640, 5), (1024, 1280, 7), (256, 1280, 7), (16384, 320, 5)]
    if len(sys.argv) > 1:  # e.g. "1024,1280,7;1024,1280,3"
        cases = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1].split(";")]
    for (M, N, tile) in cases:
        rows = []
        for K in (64, 128, 320, 640, 1280, 2560):
            x = torch.randn(S * M, K, device=dev).to(dt)
            w = (torch.randn(S, N, K, device=dev) * K ** -0.5).to(dt)
            b = torch.randn(S, N, device=dev)
            r = torch.randn(S * M, N, device=dev).to(dt)
            us = time_graph(lambda: ops.linear(x, w, b, res=r, tile=tile, splitk=1, streams=S))
            us_nores = time_graph(lambda: ops.linear(x, w, None, tile=tile, splitk=1, streams=S))
            rows.append((K, round(us, 2), round(us_nores, 2)))
        (k0, t0, _), (k1, t1, _) = rows[0], rows[-1]
        slope = (t1 - t0) / ((k1 - k0) / 64)
        print(json.dumps(dict(M=M, N=N, tile=tile, us_by_K=rows, per_chunk_us=round(slope, 3),
                              fixed_us=round(t0 - slope, 2))), flush=True)


if __name__ == "__main__":
    main()
