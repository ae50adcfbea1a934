#!/usr/bin/env python3
"""Isolated timing of the weight-streaming conv (csrc/wsconv.hip) against the tuned LDS-tiled implicit GEMM on the
resnet conv shapes of the benchmarked step (cfg 3, B = 4, two grouped streams).

    python tools/wsconv_bench.py [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import ops  # noqa: E402
from uni_renderer_amd.layers import pack_conv3x3, pack_matrix  # noqa: E402

# (H = W, Cin, N, tail channels (t0, t1), hilo residual)
SHAPES = [(64, 320, 320, (0, 0), True), (64, 640, 320, (0, 0), False), (64, 320, 320, (320, 320), True),
          (64, 960, 320, (0, 0), False), (64, 320, 320, (640, 320), True),
          (32, 320, 640, (0, 0), False), (32, 640, 640, (320, 0), True), (32, 640, 640, (0, 0), True),
          (32, 1280, 640, (0, 0), False), (32, 1920, 640, (0, 0), False), (32, 640, 640, (1280, 640), True), (32, 960, 640, (0, 0), False),
          (16, 640, 1280, (0, 0), False), (16, 1280, 1280, (640, 0), True), (16, 1280, 1280, (0, 0), True),
          (16, 2560, 1280, (0, 0), False), (16, 1280, 1280, (1280, 1280), True), (16, 1920, 1280, (0, 0), False)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--only", default="", help="comma list of shape indices")
    ap.add_argument("--sk", default="", help="comma list of split-K values to try (default: policy, x2, /2)")
    a = ap.parse_args()
    dt, S, B = torch.float16, 2, a.batch
    tot0 = tot1 = 0.0
    shapes = [SHAPES[int(i)] for i in a.only.split(",")] if a.only else SHAPES
    for (H, C, N, (ct0, ct1), hilo) in shapes:
        x = torch.randn(S * B, H, H, C, device="cuda").to(dt)
        cb = ops.conv_cblock(C)
        wt = [torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5 for _ in range(S)]
        w = torch.stack([pack_conv3x3(t, dt, cblock=cb) for t in wt])
        tail = None
        if ct0:
            t0 = torch.randn(S * B, H, H, ct0, device="cuda").to(dt)
            t1 = torch.randn(S * B, H, H, ct1, device="cuda").to(dt) if ct1 else None
            wtl = torch.stack([pack_matrix(torch.randn(N, ct0 + ct1, device="cuda") * (ct0 + ct1) ** -0.5, dt) for _ in range(S)])
            w = torch.cat([w, wtl], 2).contiguous()
            tail = (t0, t1)
        ws = torch.stack([ops.wsconv_images(w[i]) for i in range(S)])
        bias = torch.randn(S, N, device="cuda")
        kw = dict(streams=S, cblock=cb, tail=tail, hilo=hilo)
        K = w.shape[-1]
        M = B * H * H
        y0 = ops.conv3x3(x, w, bias, **kw)
        t_ig = timeit(lambda: ops.conv3x3(x, w, bias, **kw), a.iters)
        sk0 = ops.wsconv_splitk(M, N, K, S)
        cands = [int(v) for v in a.sk.split(",")] if a.sk else sorted({max(1, sk0 // 2), sk0, sk0 * 2})
        res = []
        for sk in cands:
            if sk > K // 320:
                continue
            y1 = ops.conv3x3(x, w, bias, ws=ws, splitk=sk, **kw)
            err = float((y1.float() - y0.float()).abs().max() / y0.float().abs().max())
            res.append((timeit(lambda: ops.conv3x3(x, w, bias, ws=ws, splitk=sk, **kw), a.iters), sk, err))
        best = min(res)
        fl = 2.0 * S * M * N * K
        print(f"H{H} C{C} N{N} tail{ct0}+{ct1} K{K}: igemm {t_ig:7.1f} us ({fl / t_ig / 1e6:6.0f} TF) | ws "
              + "  ".join(f"sk{sk}: {t:7.1f} us (err {e:.1e})" for t, sk, e in res)
              + f" | best {best[0]:7.1f} us ({fl / best[0] / 1e6:6.0f} TF) x{t_ig / best[0]:.2f}", flush=True)
        tot0 += t_ig
        tot1 += min(best[0], t_ig)
    print(f"sum: igemm {tot0:.0f} us, best-of {tot1:.0f} us")


if __name__ == "__main__":
    main()
