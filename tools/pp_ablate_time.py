#!/usr/bin/env python3
"""Time one ping-pong tile on a few conv / GEMM problems (isolated, graph replay) -- run once per ablation build of
csrc/igemm_pp.hip (UR_LIB_PATH=gpurun_ab/liburhip_ppabl<bits>.so; results of ablated builds are meaningless numerically)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from uni_renderer_amd import ops  # noqa: E402
from uni_renderer_amd.layers import pack_conv3x3  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[1]


def main():
    tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "9,49").split(",")]
    dt, S, B = torch.float16, 2, 4
    out = []
    for (H, C, N) in [(64, 320, 320), (64, 640, 320), (32, 640, 640)]:
        x = torch.randn(S * B, H, H, C, device="cuda").to(dt)
        cb = int(os.environ.get("UR_TEST_CBLOCK", "0")) or ops.conv_cblock(C)
        w = torch.stack([pack_conv3x3(torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5, dt, cblock=cb) for _ in range(S)])
        bias = torch.randn(S, N, device="cuda")
        for tile in tiles:
            sk = 1 if H == 64 else 2
            t = timeit(lambda: ops.conv3x3(x, w, bias, streams=S, cblock=cb, tile=tile, splitk=sk))
            fl = 2.0 * S * B * H * H * N * 9 * C
            out.append(f"conv H{H} C{C} N{N} tile {tile} sk{sk}: {t:7.1f} us ({fl / t / 1e6:6.0f} TF)")
    x = torch.randn(S, 4096, 640, device="cuda").to(dt)
    w = torch.randn(S, 5120, 640, device="cuda").to(dt) * 0.04
    for tile in tiles:
        t = timeit(lambda: ops.linear(x, w, None, tile=tile, splitk=1, streams=S))
        out.append(f"gemm M4096 N5120 K640 z2 tile {tile}: {t:7.1f} us ({2.0 * S * 4096 * 5120 * 640 / t / 1e6:6.0f} TF)")
    print(os.environ.get("UR_LIB_PATH", "default"), "cblock", os.environ.get("UR_TEST_CBLOCK", "policy"), *out, sep="\n  ", flush=True)


if __name__ == "__main__":
    main()
