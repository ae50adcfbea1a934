#!/usr/bin/env python3
"""Idle time between the kernels of the graphed step: reads a rocprofv3 --kernel-trace CSV, takes the steady-state part of
the run (the last N graph replays = the timed region of bench.py), and reports per step: wall, sum of kernel durations,
total gap, and the gap histogram by the kernel that FOLLOWS the gap.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
    python tools/gap_analysis.py gpurun_out/trace/*/*_kernel_trace.csv --steps 10
"""
import argparse
import collections
import csv
import glob
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    rows = []
    for pat in a.csv:
        for path in glob.glob(pat):
            with open(path) as f:
                for r in csv.DictReader(f):
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the timed region: the last `steps` repetitions of the most frequent launch sequence length
    names = [r[2] for r in rows]
    # find the period: the step's first kernel is the one that follows the largest gaps regularly; use the last kernel
    # count per step = (count of a kernel that runs once per step) -- take the most common count among rare kernels
    cnt = collections.Counter(names)
    once = [k for k, v in cnt.items() if a.steps <= v <= a.steps + 8]
    if not once:
        raise SystemExit("no once-per-step kernel found")
    marker = once[0]
    idx = [i for i, n in enumerate(names) if n == marker][-a.steps - 1:]
    per = idx[-1] - idx[-2]
    res = []
    for s0, s1 in zip(idx[:-1], idx[1:]):
        seg = rows[s0:s1 + 1]  # marker .. next marker: one period
        wall = seg[-1][0] - seg[0][0]
        busy = sum(e - s for s, e, _ in seg[:-1])
        res.append((wall, busy, len(seg) - 1))
    seg = rows[idx[-2]:idx[-1] + 1]
    gaps = collections.defaultdict(lambda: [0, 0])
    big = []
    for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
        g = s1 - e0
        key = n1.split("(")[0][-50:]
        gaps[key][0] += 1
        gaps[key][1] += g
        big.append((g, n0[-40:], n1[-40:]))
    wall = sum(r[0] for r in res) / len(res)
    busy = sum(r[1] for r in res) / len(res)
    out = dict(kernels_per_step=res[-1][2], wall_us=round(wall / 1e3, 1), busy_us=round(busy / 1e3, 1),
               gap_us=round((wall - busy) / 1e3, 1), gap_frac=round((wall - busy) / wall, 4), period_launches=per)
    print(json.dumps(out))
    print("gap by following kernel (count, total us, mean us):")
    for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {n:5d} {t / 1e3:9.1f} {t / n / 1e3:7.2f}  {k}")
    print("largest single gaps:")
    for g, a0, a1 in sorted(big, reverse=True)[:10]:
        print(f"  {g / 1e3:8.2f} us  {a0}  ->  {a1}")


if __name__ == "__main__":
    main()
