#!/bin/bash
# round 4, run 29: W^T of the Linear weights in multi-tensor launches at cast time; where the small elementwise launches of a
# training step come from
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -x -q -k "train or transpose" 2>&1 | tail -3
for i in 1 2; do
  for f in "UR_BATCH_WT=1" "UR_BATCH_WT=0"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
  done
done
python tools/find_fills.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/find_fills.txt; cat gpurun_out/r04/find_fills.txt
