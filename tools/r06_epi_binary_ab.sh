#!/bin/bash
# Does the epilogue-statistics code slow the igemm kernels when it is NOT used?  The tree before the change (gpurun_ab/pre_epi) against
# the current tree with the feature off and on, alternating on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { (cd $1 && UR_EXPERIMENT=$2 timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
for rep in 1 2 3; do
  echo "pre-change tree: $(run $R/gpurun_ab/pre_epi '')"
  echo "current, no_epi_gn_stats: $(run $R no_epi_gn_stats)"
  echo "current, epi_gn_stats: $(run $R epi_gn_stats)"
done
