#!/bin/bash
# Library-against-library A/B on one box: the tree's liburhip.so and every gpurun_ab/liburhip_*.so, alternating, full step + cfg 2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { (cd $R && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
run2() { (cd $R && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 200 --direction render --batch 2 --latent 32 --dtype bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); }
for rep in 1 2 3; do
  echo "tree: $(run X=1)   cfg2 $(run2 X=1)"
  for lib in $R/gpurun_ab/liburhip_*.so; do
    echo "$(basename $lib): $(run UR_LIB_PATH=$lib)   cfg2 $(run2 UR_LIB_PATH=$lib)"
  done
done
