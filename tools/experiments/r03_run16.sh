#!/bin/bash
# same-box A/B: committed HEAD (gpurun_ab/, exported before the run) vs the working tree, graphed training step
for rep in 1 2; do
  echo "== A (HEAD)"; (cd gpurun_ab && timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "== B (tree)"; timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
