#!/usr/bin/env python3
"""Ping-pong tiles (csrc/igemm_pp.hip) against the shipped tuning table, problem by problem, on the implicit-GEMM problems
of the benchmarked step (cfg 3, grouped executor): isolated graph-replay timing (tools/tune_igemm.py's method) of the
table's (tile, split-K) and of every ping-pong tile at split-K 1 / 2 / 4 / 8, plus the rel-L2 of the ping-pong output
against the table's output on the same operands.

    python tools/pp_ab.py [--min-us 25] [--out gpurun_out/r04/pp_ab.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--min-us", type=float, default=25.0, help="skip problems whose table plan runs faster than this")
    ap.add_argument("--tiles", default="49,50,51,52,53,54,55")
    ap.add_argument("--max-wgs", type=int, default=1024, help="skip candidates with more workgroups than this (512 with split-K)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04", "pp_ab.json"))
    a = ap.parse_args()
    import bench
    import tune_igemm as T
    from uni_renderer_amd import ops

    dev = torch.device("cuda:0")
    dt = torch.float16
    models = bench.build_models(dev, dt)
    calls = T.collect(models, bench.make_inputs(a.batch, a.latent, dev, dt, seed=7), grouped=True)
    tiles = [int(t) for t in a.tiles.split(",")]
    rows, tot_base, tot_best = [], 0.0, 0.0
    for key, kw in sorted(calls.items()):
        M, N, K, taps, zb = key
        plan = ops.plan_igemm(M, N, K, taps, zb)
        t_base = T.time_cfg(kw, plan[0], plan[1])
        if t_base is None or t_base * 1e3 < a.min_us:
            continue
        out = kw["out"]
        ops.igemm(**dict(kw, tile=plan[0], splitk=plan[1]))
        ref = out.float().clone()
        res = {}
        for tile in tiles:
            bm, bn = ops._TILES[tile][:2]
            for sk in (1, 2, 4, 8):
                if sk > 1 and K // 64 < 4 * sk:
                    continue
                wgs = -(-M // bm) * -(-N // bn) * zb * sk
                if wgs > a.max_wgs or (sk > 1 and wgs > a.max_wgs // 2):
                    continue
                t = T.time_cfg(kw, tile, sk)
                if t is None:
                    continue
                ops.igemm(**dict(kw, tile=tile, splitk=sk))
                err = float((out.float() - ref).norm() / ref.norm())
                res[f"{tile},{sk}"] = (round(t * 1e3, 2), err)
        if not res:
            continue
        best = min(res, key=lambda k: res[k][0])
        fl = 2.0 * M * N * K * zb
        rows.append(dict(M=M, N=N, K=K, taps=taps, z=zb, table=list(plan), table_us=round(t_base * 1e3, 2),
                         table_tf=round(fl / t_base / 1e9, 1), best_pp=best, best_pp_us=res[best][0],
                         best_pp_tf=round(fl / res[best][0] / 1e6, 1), max_err=max(v[1] for v in res.values()), all=res))
        tot_base += t_base * 1e3
        tot_best += min(t_base * 1e3, res[best][0])
        print(f"M={M:6d} N={N:5d} K={K:6d} taps={taps} z={zb}: table {plan} {t_base * 1e3:8.2f} us ({fl / t_base / 1e9:6.0f} TF) | "
              f"pp best {best:>5s} {res[best][0]:8.2f} us ({fl / res[best][0] / 1e6:6.0f} TF) x{t_base * 1e3 / res[best][0]:.2f} "
              f"err {max(v[1] for v in res.values()):.1e} | " + " ".join(f"{k}:{v[0]:.1f}" for k, v in sorted(res.items())), flush=True)
    print(f"sum over {len(rows)} distinct problems (one launch each): table {tot_base:.0f} us, best-of {tot_best:.0f} us")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"))


if __name__ == "__main__":
    main()
