#!/bin/bash
for rep in 1 2 3; do
  echo "== A (HEAD)"; (cd gpurun_ab && timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "== B (tree)"; timeout 600 python tools/train_bench.py --steps 5 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
timeout 600 python -m pytest tests/test_optim_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -2
