#!/bin/bash
# Did the group-blocked-slab code in igemm_kernel's split-K epilogue (round 6, used by UR_EXPERIMENT=splitk_gn only) slow the kernels
# that do not use it?  Same tree, library with igemm.hip of the commit before it (gpurun_ab/liburhip_preslab.so) vs the current one.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { (cd $R && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
for rep in 1 2 3; do
  echo "igemm.hip before the slab code: $(run UR_LIB_PATH=$R/gpurun_ab/liburhip_preslab.so)"
  echo "current library: $(run X=1)"
done
