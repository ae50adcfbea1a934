#!/bin/bash
# library (tree, raster auto) vs the same library with n fastest everywhere vs the previous commit's library, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
one() { UR_EXPERIMENT=$1 timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 ${@:2} 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
{
for rep in 1 2 3 4; do
  echo "cfg3 [new lib, auto] $(one "")"
  echo "cfg3 [new lib, n fastest] $(one "igemm_raster=1")"
  echo "cfg3 [prev lib] $(UR_LIB_PATH=$PREV_LIB one "")"
done
} > $O/r06_raster_ab2.txt 2>&1
cat $O/r06_raster_ab2.txt
