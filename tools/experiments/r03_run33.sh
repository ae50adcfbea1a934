#!/bin/bash
# same-box A/B of the graphed training step: d24db32 (before this round's training changes) vs the tree
for rep in 1 2 3; do
  echo "== A (d24db32)"; (cd gpurun_ab && timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "== B (tree)"; timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
