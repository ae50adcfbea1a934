#!/usr/bin/env python3
"""GroupNorm(+SiLU) timing by level (graph replay of 20 calls, z = 2 grouped shapes of the headline step)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ab_gemm import time_graph  # noqa: E402
from uni_renderer_amd import ops  # noqa: E402


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    hilo = os.environ.get("GN_LO", "1") != "0"
    for (hw, c0, c1) in [(64, 320, 0), (64, 320, 320), (64, 640, 320), (32, 640, 0), (32, 320, 0), (32, 640, 640), (32, 1280, 640), (16, 1280, 0),
                         (16, 640, 0), (16, 1280, 640), (16, 1280, 1280), (8, 1280, 0), (8, 1280, 1280)]:
        B = 8
        x = torch.randn(B, hw, hw, c0, device=dev).to(dt)
        if hilo:
            x.lo = ops.lo_encode(torch.randn(B, hw, hw, c0, device=dev) * 1e-4, dt)
        x1 = torch.randn(B, hw, hw, c1, device=dev).to(dt) if c1 else None
        C = c0 + c1
        g, b = torch.randn(2 * C, device=dev), torch.randn(2 * C, device=dev)
        us = time_graph(lambda: ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, fused=False))
        usf = time_graph(lambda: ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, fused=True, resident=False))
        usr = time_graph(lambda: ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, fused=True, resident=True))
        res_fit = ops.gn_resident_fits(hw * hw, c0, c1, 32, dt)
        a = ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, fused=False).float()
        f = ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, fused=True).float()
        if os.environ.get("GN_SWEEP"):
            ns0, na0 = ops._gn_chunks_bytes(B, hw * hw, C, 2)
            res = {}
            for ns in sorted({max(1, ns0 // 2), ns0, min(32, ns0 * 2)}):
                for na in sorted({max(1, na0 // 4), max(1, na0 // 2), na0, na0 * 2}):
                    res[f"{ns},{na}"] = round(time_graph(lambda: ops.groupnorm(x, g, b, 1e-5, x1=x1, silu=True, streams=2, nstat=ns, napply=na)), 2)
            print(json.dumps(dict(hw=hw, c0=c0, c1=c1, default=f"{ns0},{na0}", sweep=dict(sorted(res.items(), key=lambda kv: kv[1])[:5]))), flush=True)
        print(json.dumps(dict(hw=hw, c0=c0, c1=c1, hi_MB=round(B * hw * hw * C * 2 / 1e6, 1), two_launch_us=round(us, 2),
                              fused_two_sweep_us=round(usf, 2), fused_resident_us=(round(usr, 2) if res_fit else None),
                              max_abs_diff=float((a - f).abs().max()))), flush=True)


if __name__ == "__main__":
    main()
