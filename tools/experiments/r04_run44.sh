#!/bin/bash
# round 4, run 43: gamma / beta gradients of all LayerNorms / GroupNorms of a network summed in one launch at the parameter barrier
# (backward.NormSums, ur_colsum_multi): parity of the deferred step against the immediate one, same-box A/B of the graphed step
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -k "deferred or graph or oracle_autograd" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_backward_gpu.py -x -q -k "norm" 2>&1 | tail -2
for i in 1 2; do
  for f in "UR_NORM_DEFER=1" "UR_NORM_DEFER=0"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160
  done
done
