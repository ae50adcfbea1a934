#!/bin/bash
# same-box A/B of the inference step: the commit of the first round-3 collection (128c125, 11.67 ms there) vs the tree
for rep in 1 2 3; do
  echo "== A (128c125)"; (cd gpurun_ab && timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*')
  echo "== B (tree)"; timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
