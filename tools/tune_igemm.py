#!/usr/bin/env python3
"""Measure, on a real MI355X, the best (tile, split-K) of ur_igemm for every distinct implicit-GEMM problem of a
dual-stream step and write uni_renderer_amd/igemm_tuning.json (read by ops.plan_igemm).

    python tools/tune_igemm.py [--batch 4] [--latent 64] [--dtype fp16] [--also "2,32;1,128"]

One eager step is run with ops.igemm intercepted to collect the call arguments (tensors kept alive); each unique
(M, N, K, taps, zbatch) is then re-launched with every candidate configuration, timed with HIP events on the
launch stream.  Default timing: the launch is captured 8x into a HIP graph and replayed (median of 3 replays) -- the
regime the product runs in, free of host pacing; ``--eager`` times plain back-to-back launches instead.
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(models, inputs, grouped=False, run_decoder=True):
    from uni_renderer_amd import ops
    from uni_renderer_amd.fused import GroupedDualStreamStep
    from uni_renderer_amd.graph import dual_stream_step

    calls = {}
    orig = ops.igemm

    def spy(**kw):
        key = (kw["M"], kw["N"], kw["K"], kw.get("taps", 1), kw.get("zbatch", 1))
        if key not in calls:
            calls[key] = dict(kw)
        return orig(**kw)

    ops.igemm = spy
    try:
        with torch.no_grad():
            if grouped:
                GroupedDualStreamStep(*models)(*inputs, run_decoder=run_decoder)
            else:
                dual_stream_step(*models, *inputs, run_decoder=run_decoder)
        torch.cuda.synchronize()
    finally:
        ops.igemm = orig
    return calls


def collect_train(B, L, dev, dtype):
    """igemm problems of one training step (forward + backward, tools/train_bench.py's workload)."""
    import bench
    from uni_renderer_amd import ops
    from uni_renderer_amd.train_step import train_step

    nets = bench.build_models(dev, torch.float32)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    calls = {}
    orig = ops.igemm

    def spy(**kw):
        key = (kw["M"], kw["N"], kw["K"], kw.get("taps", 1), kw.get("zbatch", 1))
        if key not in calls:
            calls[key] = dict(kw)
        return orig(**kw)

    ops.igemm = spy
    try:
        train_step(nets, batch, dtype=dtype)
        torch.cuda.synchronize()
    finally:
        ops.igemm = orig
    for m in nets:
        m.zero_grad(set_to_none=True)
    return calls


_side = None
_scrub = None        # --scrub: buffers streamed between two timed launches (evicts the XCDs' L2s)
_scrub_ms = None     # time of the scrub pass alone (subtracted)


def _scrub_pass():
    # read + write 2 x 48 MB: more than the 32 MB of L2 on the chip, a fraction of the 256 MB Infinity Cache -- what a
    # launch finds IN SITU: its activations were written by another kernel (Infinity Cache / a different XCD's L2), its
    # weights were last read a whole step ago
    _scrub[0].add_(_scrub[1])


def time_cfg(kw, tile, splitk, rounds=3, iters=8, graph=True):
    from uni_renderer_amd import ops

    global _side, _scrub_ms
    kw = dict(kw)
    kw["tile"], kw["splitk"] = tile, splitk
    try:
        ops.igemm(**kw)  # warm (also sets the LDS attribute)
        torch.cuda.synchronize()
    except RuntimeError:
        return None
    ts = []
    if graph:
        if _side is None:
            _side = torch.cuda.Stream()
        if _scrub is not None and _scrub_ms is None:
            gs = torch.cuda.CUDAGraph()
            with torch.cuda.stream(_side):
                with torch.cuda.graph(gs, stream=_side):
                    for _ in range(iters):
                        _scrub_pass()
            torch.cuda.synchronize()
            gs.replay()
            torch.cuda.synchronize()
            tt = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gs.replay()
                e1.record()
                torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1) / iters)
            _scrub_ms = statistics.median(tt)
            print(f"[tune] scrub pass alone: {_scrub_ms * 1e3:.1f} us", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(_side):
            with torch.cuda.graph(g, stream=_side):
                for _ in range(iters):
                    if _scrub is not None:
                        _scrub_pass()
                    ops.igemm(**kw)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters - (_scrub_ms or 0.0))
        del g
        return statistics.median(ts)
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.igemm(**kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--also", default="", help='extra "batch,latent" pairs separated by ;')
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of graph replays")
    ap.add_argument("--only-missing", action="store_true", help="tune only problems absent from the existing table")
    ap.add_argument("--train", action="store_true", help="tune the problems of a training step (forward + backward)")
    ap.add_argument("--tiles", default="", help="comma-separated candidate tile ids (default: all)")
    ap.add_argument("--scrub", action="store_true", help="evict the L2s between timed launches (the in-situ regime: operands "
                    "come from the Infinity Cache / HBM, not from a previous replay's L2 lines)")
    ap.add_argument("--out", default=os.path.join(ROOT, "uni_renderer_amd", "igemm_tuning.json"))
    ap.add_argument("--report", default=os.path.join(ROOT, "gpurun_out", "tune_report.json"))
    args = ap.parse_args()
    import bench
    from uni_renderer_amd import ops

    dev = torch.device("cuda:0")
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.scrub:
        global _scrub
        _scrub = (torch.zeros(24 << 20, dtype=torch.float16, device=dev), torch.zeros(24 << 20, dtype=torch.float16, device=dev))
    models = None if args.train else bench.build_models(dev, dtype)
    tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else list(ops._TILES)
    shapes = [(args.batch, args.latent)] + [tuple(int(v) for v in p.split(",")) for p in args.also.split(";") if p]
    ops.load_tuning_table("/nonexistent")  # start from the analytic planner
    table, report = {}, []
    seed_table = os.path.join(ROOT, "uni_renderer_amd", "igemm_tuning.json")
    for pth in (seed_table, args.out):
        if os.path.exists(pth):
            table.update(json.load(open(pth)))
    for (B, L) in shapes:
        calls = collect_train(B, L, dev, dtype) if args.train else collect(models, bench.make_inputs(B, L, dev, dtype, seed=7))
        for grouped, dec in (() if args.train else ((True, True), (True, False))):  # the rendering direction runs the up path ungrouped (z = 1)
            extra = collect(models, bench.make_inputs(B, L, dev, dtype, seed=7), grouped=grouped, run_decoder=dec)
            calls.update({k: v for k, v in extra.items() if k not in calls})
        if args.only_missing:
            calls = {k: v for k, v in calls.items() if f"{k[0]},{k[1]},{k[2]},{k[3]},{k[4]}" not in table}
        print(f"[tune] batch {B} latent {L}: {len(calls)} distinct problems", flush=True)
        for key, kw in sorted(calls.items()):
            M, N, K, taps, zb = key
            res = {}
            for tile in tiles:
                for sk in (1, 2, 4, 8, 16):
                    if sk > 1 and (zb > 4 or K // 64 < 4 * sk):
                        continue
                    t = time_cfg(kw, tile, sk, graph=not args.eager)
                    if t is not None:
                        res[(tile, sk)] = t
            best = min(res, key=res.get)
            default = ops.plan_igemm(M, N, K, taps, zb)
            fl = 2.0 * M * N * K * zb
            table[f"{M},{N},{K},{taps},{zb}"] = list(best)
            report.append(dict(M=M, N=N, K=K, taps=taps, z=zb, best=list(best), best_us=round(res[best] * 1e3, 2),
                               best_tflops=round(fl / res[best] / 1e9, 1), default=list(default),
                               default_us=round(res.get(tuple(default), float("nan")) * 1e3, 2),
                               all={f"{t},{s}": round(v * 1e3, 2) for (t, s), v in sorted(res.items())}))
            print(f"  M={M:6d} N={N:5d} K={K:6d} taps={taps} z={zb}: best tile {best[0]} splitk {best[1]} "
                  f"{res[best] * 1e3:8.2f} us ({fl / res[best] / 1e9:7.1f} TF/s)  planner {default} "
                  f"{res.get(tuple(default), float('nan')) * 1e3:8.2f} us", flush=True)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    os.makedirs(os.path.dirname(args.report), exist_ok=True)
    with open(args.report, "w") as f:
        json.dump(report, f)
    print(f"[tune] wrote {len(table)} entries to {args.out}")


if __name__ == "__main__":
    main()
