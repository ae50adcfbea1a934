#!/usr/bin/env python3
"""Per-kernel floors inside a replayed HIP graph: a dependent chain of trivial kernels (ur_add), small LayerNorms, and the
smallest GEMM problems of the step with and without their epilogue operands -- what a launch costs before it does any work,
what a K chunk adds, what the epilogue adds.  Evidence file: profiles/r05_launch_floor.txt."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import ops  # noqa: E402


def chain(fn, n=200, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / reps / n


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    res = {}
    for numel in (8192, 1 << 20, 1 << 23):
        a = [torch.randn(numel, device=dev).to(dt)]
        b = torch.randn(numel, device=dev).to(dt)

        def add():
            a[0] = ops.add(a[0], b)
        res[f"dependent_add_{numel}_us"] = round(chain(add), 2)
    for rows, C in ((256, 1280), (4096, 640), (16384, 320)):
        x = [torch.randn(rows, C, device=dev).to(dt)]
        g_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)

        def ln():
            x[0] = ops.layernorm(x[0], g_, b_)
        res[f"dependent_layernorm_{rows}x{C}_us"] = round(chain(ln), 2)
    for (M, N, K) in ((1024, 1280, 1280), (1024, 1280, 64), (4096, 640, 640), (4096, 640, 64), (256, 1280, 1280)):
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(dt)
        r.lo = torch.zeros(M, N, dtype=torch.uint8, device=dev)
        for name, fn in (("plain", lambda: ops.linear(x, w)), ("bias", lambda: ops.linear(x, w, b)),
                         ("bias_res", lambda: ops.linear(x, w, b, res=r, res_lo=None)),
                         ("bias_res_hilo", lambda: ops.linear(x, w, b, res=r, hilo=True))):
            res[f"gemm_{M}x{N}x{K}_{name}_us"] = round(chain(fn, n=100), 2)  # independent launches back to back, operands hot
    print(json.dumps(res))


if __name__ == "__main__":
    main()
