#!/bin/bash
# Kernel CODE SIZE against step time (round 6): the library built from the tree (current igemm epilogue) vs gpurun_ab/liburhip_base.so
# (the epilogue before the change under test), alternating on one box; full step, cfg 2 and the hoisted step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { (cd $R && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
run2() { (cd $R && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 200 --direction render --batch 2 --latent 32 --dtype bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
for rep in 1 2 3; do
  echo "base library: $(run UR_LIB_PATH=$R/gpurun_ab/liburhip_base.so)   cfg2 $(run2 UR_LIB_PATH=$R/gpurun_ab/liburhip_base.so)"
  echo "tree library: $(run X=1)   cfg2 $(run2 X=1)"
done
echo "hoisted base: $(cd $R && UR_LIB_PATH=$R/gpurun_ab/liburhip_base.so timeout 300 python tools/hoist_bench.py 2>/dev/null | tail -1 | cut -c1-120)"
echo "hoisted tree: $(cd $R && timeout 300 python tools/hoist_bench.py 2>/dev/null | tail -1 | cut -c1-120)"
