#!/bin/bash
# Register-resident one-launch GroupNorm (round 6): kernel-level table and in-step A/B, one box.
cd ${GRAFT_REPO_ROOT:-.}
timeout 300 python tools/gn_bench.py 2>/dev/null
run() { UR_EXPERIMENT=$1 timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
  echo "no_gn_resident: $(run no_gn_resident)"
  echo "gn_resident (<= 1024 rows): $(run gn_resident)"
  echo "gn_resident + 64x64 level (gn_resident_max_rows=4096): $(run gn_resident_max_rows=4096)"
done
