#!/usr/bin/env python3
"""Time the attention backward of the three transformer levels of cfg 4 (B=4, 8 heads): flash (ur_attention_backward)
vs the materialised-P path, HIP events around 10 calls each.  -> gpurun_out/attn_bwd_bench.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from uni_renderer_amd import backward as bw
    dev = torch.device("cuda:0")
    rows = []
    for dtype in (torch.bfloat16,):
        for (B, H, d, T) in ((4, 8, 40, 4096), (4, 8, 80, 1024), (4, 8, 160, 256), (4, 8, 160, 64)):
            g = torch.Generator(device="cpu").manual_seed(1)
            mk = lambda: (torch.randn(B, T, H * d, generator=g) * 0.5).to(dtype).to(dev)
            q, k, v, do = mk(), mk(), mk(), mk()
            # the forward kernel's output and row log-sum-exp, as autograd_ops.Attention hands them to the backward
            from uni_renderer_amd import ops
            stats = bw.flash_stats(B, H, T, T, d, dev)
            vt = bw._pad_rows64(bw.transpose2d(v))
            o = ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=T, d=d, ldq=H * d, ldk=H * d, lse=stats[0])
            res = {}
            for name, kw in (("flash", dict(o=o, stats=stats)), ("materialised", dict())):
                for _ in range(2):
                    bw.attention_backward(q, k, v, do, H, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    bw.attention_backward(q, k, v, do, H, **kw)
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) / 10 * 1e3
            fl = 4.0 * T * T * d * B * H * 2.5  # 2.5 x the forward's 4 T^2 d per slice
            rows.append(dict(B=B, H=H, d=d, T=T, dtype=str(dtype), flash_us=round(res["flash"], 1),
                             materialised_us=round(res["materialised"], 1),
                             flash_tflops_useful=round(fl / res["flash"] / 1e6, 1)))
            print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_bwd_bench.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
