#!/bin/bash
# round 4, run 35: GroupNorm backward of the small maps in one launch: parity, training tests, same-box A/B of the graphed step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "groupnorm" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
  for f in "UR_GN_BWD_FUSED_MAX_ROWS=1024" "UR_GN_BWD_FUSED_MAX_ROWS=0" "UR_GN_BWD_FUSED_MAX_ROWS=4096"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
  done
done
