#!/usr/bin/env python3
"""The hoisted sampling steps (uni_renderer_amd/hoist.py) at the headline shape: replay time of the prologue graph and of the
per-step graph, both directions, next to the full enc + unet + dec step; and the implicit-GEMM problems of the hoisted
graphs that have no row in the tuning table (they fall to the analytic planner).

    python tools/hoist_bench.py [--batch 4] [--latent 64] [--replays 50]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uni_renderer_amd import ops  # noqa: E402
from uni_renderer_amd.graph import GraphedDualStreamStep, GraphedHoistedStep  # noqa: E402


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--replays", type=int, default=50)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--shape-table", default="", help="write per-(kernel, problem shape) event tables of the hoisted steps here")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    models = bench.build_models(dev, dt)
    inputs = bench.make_inputs(args.batch, args.latent, dev, dt, seed=100)
    res = {"batch": args.batch, "latent": args.latent, "dtype": args.dtype}
    calls, missing = {}, {}
    orig = ops.plan_igemm

    def spy(M, N, K, taps=1, zbatch=1):
        k = f"{M},{N},{K},{taps},{zbatch}"
        calls[k] = calls.get(k, 0) + 1
        tab = ops._tune_table if ops._tune_table is not None else ops.load_tuning_table()
        if k not in tab and not (ops._site and f"{k}@{ops._site}" in tab):
            missing[k] = missing.get(k, 0) + 1
        return orig(M, N, K, taps, zbatch)

    for name, run_decoder in (("inverse", True), ("render", False)):
        g = GraphedHoistedStep(*models, batch=args.batch, latent_hw=args.latent, cross_dim=768, dtype=dt, device=dev,
                               run_decoder=run_decoder)
        g.load_inputs(*inputs)
        calls.clear()
        missing.clear()
        ops.plan_igemm = spy
        g.capture(warmup=1)
        ops.plan_igemm = orig
        res[name] = dict(prologue_ms=round(timed(g.pro.replay, args.replays), 4), step_ms=round(timed(g.graph.replay, args.replays), 4),
                         untuned=dict(sorted(missing.items())))
        if args.shape_table:
            _, table, total = bench.measure_roofline(g._run, by_shape=True)
            res[name]["sum_kernel_ms_eager_step"] = round(total, 3)
            res[name]["launches"] = sum(r["calls"] for r in table)
            shapes = json.load(open(args.shape_table)) if os.path.exists(args.shape_table) else {}
            shapes[name] = table
            json.dump(shapes, open(args.shape_table, "w"))
        f = GraphedDualStreamStep(*models, batch=args.batch, latent_hw=args.latent, cross_dim=768, dtype=dt, device=dev,
                                  run_decoder=run_decoder)
        f.load_inputs(*inputs)
        f.capture(warmup=1)
        res[name]["all_networks_step_ms"] = round(timed(f.replay, args.replays), 4)
        del g, f
    print(json.dumps(res))


if __name__ == "__main__":
    main()
