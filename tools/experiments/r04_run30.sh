#!/bin/bash
# round 4, run 30: where the small elementwise launches of a training step come from; kernel statistics of the current step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python tools/find_fills.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/find_fills.txt; cat gpurun_out/r04/find_fills.txt
bash tools/experiments/r04_run28.sh > /dev/null 2>&1
