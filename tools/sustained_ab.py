#!/usr/bin/env python3
"""Burst vs sustained timing of igemm tile variants (is the isolated tuner's ranking a clock / boost artefact?).

For a few heavy problems of the step: each candidate (tile, split-K) is timed (a) as the tuner does -- an 8-launch graph,
3 replays with a host sync in between ("burst") -- and (b) sustained: the same graph replayed back to back for ~0.6 s,
time per launch over the last 0.4 s.  If the ranking changes between (a) and (b), the chip's sustained clock / current
limit, not the kernel's stall cycles, sets the step time."""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench
    import tune_igemm
    from uni_renderer_amd import ops

    dev = torch.device("cuda:0")
    models = bench.build_models(dev, torch.float16)
    calls = tune_igemm.collect(models, bench.make_inputs(4, 64, dev, torch.float16, seed=7), grouped=True)
    want = [((32768 // 2, 320, 5760, 9, 2), [(9, 1), (31, 1), (32, 1), (11, 1), (1, 1)]),
            ((16384, 320, 2880, 9, 2), [(9, 1), (31, 1), (32, 1), (5, 1)]),
            ((4096, 640, 11520, 9, 2), [(9, 2), (32, 2), (11, 2), (9, 1), (32, 1)]),
            ((1024, 1280, 11520, 9, 2), [(9, 4), (32, 4), (5, 8), (1, 8)]),
            ((16384, 2560, 320, 1, 2), [(11, 1), (9, 1), (32, 1), (10, 1)]),
            ((16384, 320, 320, 1, 2), [(9, 1), (5, 1), (32, 1), (7, 1)]),
            ((1024, 1280, 1280, 1, 2), [(7, 1), (3, 1), (5, 1), (37, 1)])]
    side = torch.cuda.Stream()
    out = []
    for key, cands in want:
        kw = calls.get(key)
        if kw is None:
            print("missing", key)
            continue
        for tile, sk in cands:
            k2 = dict(kw)
            k2["tile"], k2["splitk"] = tile, sk
            ops.igemm(**k2)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(8):
                        ops.igemm(**k2)
            torch.cuda.synchronize()
            time.sleep(0.3)  # let the chip idle like between tuner candidates
            burst = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                burst.append(e0.elapsed_time(e1) / 8)
            # sustained: ~0.2 s warm + 0.4 s timed, no host sync in between
            n_warm = max(1, int(0.2 / (burst[-1] * 8e-3)))
            n_time = max(1, int(0.4 / (burst[-1] * 8e-3)))
            for _ in range(n_warm):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_time):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            sus = e0.elapsed_time(e1) / (8 * n_time)
            row = dict(problem=key, tile=tile, splitk=sk, burst_us=round(statistics.median(burst) * 1e3, 1), sustained_us=round(sus * 1e3, 1))
            out.append(row)
            print(row, flush=True)
            del g
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sustained_ab.json"), "w"))


if __name__ == "__main__":
    main()
