#!/bin/bash
# round 4, run 42: per-head copies of the d = 40 attention backward in one launch each way: parity, same-box A/B of the graphed step
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "heads or attention or flash" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -k "graph or oracle_autograd or sd_size" 2>&1 | tail -2
for i in 1 2; do
  for f in "UR_HEADS_MULTI=1" "UR_HEADS_MULTI=0"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160
  done
done
