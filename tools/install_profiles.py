#!/usr/bin/env python3
"""Copy the summaries of an end-of-round collection (tools/experiments/r02_final.sh -> gpurun_out/) into profiles/ under a round tag.

    python tools/install_profiles.py r02

gpurun_out/ is scratch; profiles/ is what is tracked.  Copies prof_<tag>/* (collect_profiles.sh), the final_* bench /
training / loop / VAE lines, extracts every dict the ``-m gpu`` suite printed (``pytest -rP``) into
<tag>_parity_numbers.json, and regenerates profiles/README.md.
"""
import ast
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(OUT, f"prof_{tag}")
    for f in sorted(os.listdir(src)):
        if f.endswith((".json", ".csv")):
            shutil.copy(os.path.join(src, f), os.path.join(PROF, f"{tag}_{f}"))
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(PROF, "pmc_traffic.json"))  # bench.py reads this one
    for f in sorted(os.listdir(OUT)):
        if f.startswith("final_") and f.endswith((".json", ".csv", ".txt")):
            dst = os.path.join(PROF, f"{tag}_{f[len('final_'):]}")
            shutil.copy(os.path.join(OUT, f), dst)
            if f.endswith(".json"):
                # a tool's stdout can carry foreign lines (RCCL prints its version banner there): keep the JSON lines only, so that
                # every committed .json parses
                keep = []
                for line in open(dst, errors="replace"):
                    try:
                        json.loads(line)
                        keep.append(line if line.endswith("\n") else line + "\n")
                    except ValueError:
                        pass
                if keep:
                    open(dst, "w").writelines(keep)
    rows, test = [], None
    log = os.path.join(OUT, "final_pytest.log")
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            s = line.strip()
            if s.startswith("____") and s.endswith("____"):
                test = s.strip("_ ").strip()
            elif s.startswith("{") and s.endswith("}"):
                try:
                    d = json.loads(s)
                except ValueError:
                    try:
                        d = ast.literal_eval(s)
                    except (ValueError, SyntaxError):
                        continue
                if isinstance(d, dict):
                    rows.append({"test": test, **{str(k): v for k, v in d.items()}})
            elif " passed" in s and s.startswith("="):
                rows.append({"suite": s.strip("= ")})
            elif " passed" in s and "deselected" in s:
                rows.append({"suite": s})
        with open(os.path.join(PROF, f"{tag}_parity_numbers.json"), "w") as f:
            json.dump(rows, f, indent=1, default=str)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profiles_readme.py"), tag])
    print(f"installed {tag}: {len(rows)} parity rows")


if __name__ == "__main__":
    main()
