#!/usr/bin/env python3
"""(tile, slices) of ur_wgrad per weight-gradient problem of the training step -> uni_renderer_amd/wgrad_tuning.json.

One eager training step (cfg 4: B = 4, 64x64 latent, bf16; --batch / --latent for others) with backward.WGRAD_TRACE set
collects the problems "P,N,K,taps,stride"; every candidate (six tiles x slice counts 1..64) is then timed in isolation on
random operands of that shape and the fastest kept.  Problems of other configurations fall back to the library's choice.

    python tools/tune_wgrad.py [--out uni_renderer_amd/wgrad_tuning.json] [--merge] [--configs 4x64,1x64,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd import backward as B_  # noqa: E402
from uni_renderer_amd.train_step import train_step  # noqa: E402


def collect(batch_size, latent, dt):
    dev = torch.device("cuda", 0)
    nets = bench.build_models(dev, torch.float32)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    B, L = batch_size, latent
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g), t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    B_.WGRAD_TRACE, B_.wgrad_queue.trace = {}, {}
    train_step(nets, batch, optimizer=None, buckets=None, dtype=dt)
    torch.cuda.synchronize()
    seen, B_.WGRAD_TRACE = B_.WGRAD_TRACE, None
    seen.update(B_.wgrad_queue.trace)   # "P,N,K,1,0@G": G equally shaped Linear problems of one flush (backward.WgradQueue)
    B_.wgrad_queue.trace = None
    del nets
    torch.cuda.empty_cache()
    return seen


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=B_.WGRAD_TABLE_PATH)
    ap.add_argument("--merge", action="store_true", help="keep the entries of an existing table for problems not visited")
    ap.add_argument("--configs", default="4x64", help="comma-separated batch x latent pairs to collect problems from")
    ap.add_argument("--iters", type=int, default=15)
    args = ap.parse_args()
    dt = torch.bfloat16
    seen = {}
    for c in args.configs.split(","):
        b, l = (int(v) for v in c.split("x"))
        for k, n in collect(b, l, dt).items():
            seen[k] = seen.get(k, 0) + n
    print(f"[tune_wgrad] {len(seen)} problems, {sum(seen.values())} calls per step", flush=True)
    table = {}
    if args.merge and os.path.exists(args.out):
        table = {k: v for k, v in json.load(open(args.out)).items() if not k.startswith("_")}
    mk = lambda *s: torch.randn(*s, device="cuda").to(dt)
    total_best = total_auto = 0.0
    for key, calls in sorted(seen.items(), key=lambda kv: -kv[1]):
        shape, _, grp = key.partition("@")
        G = int(grp) if grp else 1
        P, N, K, taps, stride = (int(v) for v in shape.split(","))
        conv = None
        if taps == 9:
            Cc = K // 9
            # the tuner does not know the image shape: a square power-of-two map of P / 4 ... P pixels per sample reproduces it
            # for the UNet levels (B samples of H x H); B is taken from the largest square that divides P
            Bn = next(b for b in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20) if P % b == 0 and int(round((P // b) ** 0.5)) ** 2 == P // b
                      and ((P // b) & (P // b - 1)) == 0)
            Ho = int(round((P // Bn) ** 0.5))
            conv = (Ho, Ho, stride)
            mkx = lambda: mk(Bn, Ho * stride, Ho * stride, Cc)
        else:
            mkx = lambda: mk(P, K)
        if G == 1:
            dy, x = mk(P, N), mkx()
            run = lambda tile, sp: B_.wgrad(dy, x, True, conv=conv, tile=tile, splits=sp)
        else:
            items = [(mk(P, N), mkx(), torch.empty(N, K, dtype=dt, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda"))
                     for _ in range(G)]
            run = lambda tile, sp: B_.wgrad_group(items, tile=tile, splits=sp, conv=conv)
        best = None
        for tile in (1, 2, 3, 4, 5, 6):
            for sp in (1, 2, 4, 8, 16, 32, 64):
                if sp > 1 and (sp * 4 > (P + 31) // 32 or G * sp > 256):
                    continue
                t = timeit(lambda: run(tile, sp), args.iters)
                if best is None or t < best[0]:
                    best = (t, tile, sp)
        B_._wgrad_table = {}
        t_auto = timeit(lambda: run(0, 0), args.iters)
        table[key] = [best[1], best[2]]
        total_best += best[0] * calls
        total_auto += t_auto * calls
        print(f"  {key:32s} x{calls:3d}  best tile {best[1]} slices {best[2]:2d}: {best[0]:7.1f} us   (library's choice {t_auto:7.1f} us)", flush=True)
        del run
    print(f"[tune_wgrad] per step: {total_best / 1e3:.2f} ms tuned, {total_auto / 1e3:.2f} ms with the library's choice")
    table = dict(sorted(table.items()))
    table["_comment"] = ("ur_wgrad (tile, slices) per problem P,N,K,taps,stride (@G: a group of G problems in one launch) -- "
                         "tools/tune_wgrad.py, isolated timings on an MI355X")
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
