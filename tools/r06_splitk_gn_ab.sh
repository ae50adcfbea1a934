#!/bin/bash
# GroupNorm as the split-K second pass with GROUP-BLOCKED slabs (round 6) against reduce + one-launch GroupNorm, alternating on one box.
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out
for rep in 1 2; do
  for mode in "" "splitk_gn" "no_splitk_gn"; do
    [ -z "$mode" ] && tag=default || tag=$mode
    UR_EXPERIMENT=$mode timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config'].get('launches_per_step'))"
  done
done
UR_EXPERIMENT=splitk_gn timeout 300 python tools/hoist_bench.py 2>/dev/null | tail -2
UR_EXPERIMENT=no_splitk_gn timeout 300 python tools/hoist_bench.py 2>/dev/null | tail -2
