#!/usr/bin/env python3
"""Cold-weight timing of the weight-heavy implicit-GEMM launches of a dual-stream step.

tools/tune_igemm.py replays ONE launch 8x in a graph: from the second replay on its weights sit in the 256 MB Infinity
Cache.  In the real step every weight byte is cold (3.5 GB per step through a 256 MB cache), so the deep levels
(16x16 / 8x8 latents: 30-60 MB of weights per launch, a few GFLOP) are paced by HBM latency, not by what the tuner saw.

For every problem with >= --min-mb of weights this tool times, per candidate (tile, split-K):
    warm    the tuner's regime (same weights every launch)
    cold    n copies of the weights (n x bytes >= 640 MB), launch i uses copy i: every launch streams from HBM
    cold+pf cold, with ur_prefetch of copy i+1 on a second stream while launch i runs (what a weight-prefetch branch of
            the step graph would give when the prefetch is hidden under the previous layer)
and writes gpurun_out/cold_bench.json.

    python tools/cold_bench.py [--batch 4] [--latent 64] [--min-mb 6] [--tiles 2,3,5,...] [--prefetch-wgs 64]
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def graph_time(fn_list, rounds=3):
    """capture the launches of fn_list (each a callable) into one graph; median replay time per launch (ms)."""
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for f in fn_list:
                f()
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
        ts.append(e0.elapsed_time(e1) / len(fn_list))
    del g
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--min-mb", type=float, default=6.0)
    ap.add_argument("--tiles", default="")
    ap.add_argument("--splitk", default="1,2,4,8,16")
    ap.add_argument("--prefetch-wgs", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cold_bench.json"))
    args = ap.parse_args()
    import bench
    import tune_igemm
    from uni_renderer_amd import _lib, ops

    dev = torch.device("cuda:0")
    lib = _lib.load()
    models = bench.build_models(dev, torch.float16)
    calls = tune_igemm.collect(models, bench.make_inputs(args.batch, args.latent, dev, torch.float16, seed=7), grouped=True)
    tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else list(ops._TILES)
    sks = [int(s) for s in args.splitk.split(",")]
    pf_stream = torch.cuda.Stream()
    report = []
    # the prefetch kernel alone: how fast does it pull cold bytes?
    big = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(12)]
    for wgs in (16, 32, 64, 128, 256, 512):
        t = graph_time([(lambda b=b, w=wgs: lib.ur_prefetch(b.data_ptr(), b.numel(), w, torch.cuda.current_stream().cuda_stream))
                        for b in big])
        print(f"[prefetch alone] {wgs:4d} WGs: {t * 1e3:7.1f} us per 64 MB = {64 * 1.048576 / t / 1e3:6.2f} TB/s", flush=True)
        report.append(dict(prefetch_alone_wgs=wgs, us_per_64MB=round(t * 1e3, 1)))
    del big
    for key, kw in sorted(calls.items()):
        M, N, K, taps, zb = key
        w = kw["w"]
        wbytes = w.numel() * w.element_size()
        if wbytes < args.min_mb * (1 << 20):
            continue
        n = max(2, min(48, -(-(640 << 20) // wbytes)))
        copies = [w] + [w.clone() for _ in range(n - 1)]
        cur = ops.plan_igemm(M, N, K, taps, zb)
        res = {}
        for tile in tiles:
            for sk in sks:
                if sk > 1 and (zb > 4 or K // 64 < 4 * sk):
                    continue
                k2 = dict(kw)
                k2["tile"], k2["splitk"] = tile, sk
                try:
                    ops.igemm(**k2)
                    torch.cuda.synchronize()
                except RuntimeError:
                    continue

                def launch(c, k2=k2):
                    kk = dict(k2)
                    kk["w"] = c
                    ops.igemm(**kk)

                warm = graph_time([(lambda: launch(w)) for _ in range(8)])
                cold = graph_time([(lambda c=c: launch(c)) for c in copies])
                res[(tile, sk)] = (warm, cold)
        if not res:
            continue
        best_cold = min(res, key=lambda k_: res[k_][1])
        best_warm = min(res, key=lambda k_: res[k_][0])
        # prefetch variant for the table's config and the cold-best config
        pf = {}
        for cfg in {tuple(cur), best_cold}:
            if cfg not in res:
                continue
            k2 = dict(kw)
            k2["tile"], k2["splitk"] = cfg

            def step(i, k2=k2):
                main_s = torch.cuda.current_stream()
                pf_stream.wait_stream(main_s)
                nxt = copies[(i + 1) % n]
                lib.ur_prefetch(nxt.data_ptr(), wbytes, args.prefetch_wgs, pf_stream.cuda_stream)
                kk = dict(k2)
                kk["w"] = copies[i]
                ops.igemm(**kk)
                main_s.wait_stream(pf_stream)

            pf[cfg] = graph_time([(lambda i=i: step(i)) for i in range(n)])
        fl = 2.0 * M * N * K * zb
        row = dict(M=M, N=N, K=K, taps=taps, z=zb, weight_mb=round(wbytes / 2 ** 20, 1), copies=n, table=list(cur),
                   table_warm_us=round(res.get(tuple(cur), (float("nan"),) * 2)[0] * 1e3, 1),
                   table_cold_us=round(res.get(tuple(cur), (float("nan"),) * 2)[1] * 1e3, 1),
                   best_warm=list(best_warm), best_warm_us=round(res[best_warm][0] * 1e3, 1),
                   best_cold=list(best_cold), best_cold_us=round(res[best_cold][1] * 1e3, 1),
                   cold_hbm_floor_us=round(wbytes / 6.0e6, 1),
                   prefetch_us={f"{t},{s}": round(v * 1e3, 1) for (t, s), v in pf.items()},
                   all={f"{t},{s}": [round(a * 1e3, 1), round(b * 1e3, 1)] for (t, s), (a, b) in sorted(res.items())})
        report.append(row)
        print(f"M={M:6d} N={N:5d} K={K:6d} taps={taps} z={zb} W={row['weight_mb']:6.1f} MB: table {cur} warm {row['table_warm_us']:7.1f} "
              f"cold {row['table_cold_us']:7.1f} | best cold {best_cold} {row['best_cold_us']:7.1f} | +prefetch {row['prefetch_us']} "
              f"| HBM floor {row['cold_hbm_floor_us']:6.1f} us  ({fl / res[best_cold][1] / 1e9:6.0f} TF cold)", flush=True)
        del copies
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f)


if __name__ == "__main__":
    main()
