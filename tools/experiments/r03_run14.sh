#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q 2>&1 | tail -8
for f in 0 1; do
  echo "== UR_FUSED_COLSUM=$f"
  UR_FUSED_COLSUM=$f timeout 600 python tools/train_bench.py --steps 4 --graph 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -4
