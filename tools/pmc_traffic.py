#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they cannot share a pass on gfx950: TCC has 4 slots,
FETCH_SIZE takes 3 and WRITE_SIZE 2) into per-kernel-class HBM traffic per launch, keyed like bench.py's
`kernel_classes`, and write profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o p --output-format csv -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o p --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv

Units and corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what every kernel here issues, so it
is doubled; WRITE_SIZE is used as reported (uncalibrated, says the guide).
"""
import collections
import csv
import json
import os
import re
import sys

TILE = {(128, 128, 2, 2, 2): "128x128s2", (128, 64, 4, 1, 3): "128x64s3", (64, 64, 4, 1, 3): "64x64s3",
        (128, 128, 2, 2, 3): "128x128s3", (128, 64, 4, 1, 2): "128x64s2", (64, 64, 4, 1, 4): "64x64s4",
        (64, 64, 4, 1, 2): "64x64s2", (256, 128, 4, 2, 2): "256x128s2", (128, 320, 2, 5, 2): "128x320s2",
        (128, 256, 2, 4, 2): "128x256s2", (256, 256, 4, 4, 2): "256x256s2"}


def klass(name: str):
    m = re.search(r"igemm_kernelID(?:F16_|F16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(n?\d+)ELb(\d)", name)
    if m:
        bm, bn, wm, wn = (int(m.group(i)) for i in range(1, 5))
        st = m.group(5)
        st = -int(st[1:]) if st.startswith("n") else int(st)
        t = TILE.get((bm, bn, wm, wn, st), f"{bm}x{bn}s{st}w{wm * wn}")
        return f"igemm_{t}_{'conv3x3' if m.group(6) == '1' else 'gemm'}"
    m = re.search(r"attention(?:32)?_kernelID(?:F16_|F16b)Li(\d+)", name)
    if m:
        return f"attention_d{m.group(1)}"
    for k, v in (("gn_stats_kernel", "gn_stats"), ("gn_apply_kernel", "gn_apply"), ("layernorm_kernel", "layernorm"),
                 ("add_kernel", "add"), ("igemm_splitk_reduce", "igemm_splitk_reduce")):
        if k in name:
            return v
    return None


def per_class(path, counter):
    """{class: [sum, launches]}.  A split-K launch shares its kernel symbol with the plain launches of the tile; it is
    recognised by the ``igemm_splitk_reduce`` dispatch that immediately follows it (Dispatch_Id order) and filed under
    ``<class>_splitk`` -- the same split bench.py's ``kernel_classes`` make."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    agg = collections.defaultdict(lambda: [0.0, 0])
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        names = {int(r["Dispatch_Id"]): r["Kernel_Name"] for r in rows}
    else:
        names = {}
    for r in rows:
        k = klass(r["Kernel_Name"])
        if not k:
            continue
        if names and k.startswith("igemm_") and k != "igemm_splitk_reduce" and \
                "igemm_splitk_reduce" in names.get(int(r["Dispatch_Id"]) + 1, ""):
            k += "_splitk"
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    return agg


def main():
    fetch, write = per_class(sys.argv[1], "FETCH_SIZE"), per_class(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [0.0, 1])
        w = write.get(k, [0.0, 1])
        rd = 2.0 * 1024.0 * f[0] / max(f[1], 1)  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        wr = 1024.0 * w[0] / max(w[1], 1)
        out[k] = dict(hbm_bytes_per_launch=round(rd + wr), read_bytes=round(rd), write_bytes=round(wr), launches=f[1],
                      note="FETCH_SIZE*2*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section); <class>_splitk = the "
                           "launches of that kernel symbol followed by igemm_splitk_reduce (fp32 partial slabs)")
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if len(sys.argv) > 3:
        dst = sys.argv[3]
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(k, v["hbm_bytes_per_launch"], v["launches"])


if __name__ == "__main__":
    main()
