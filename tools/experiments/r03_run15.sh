#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/train_bench.py --steps 4 --graph 2>&1 | tail -1
timeout 600 python tools/train_bench.py --steps 4 2>&1 | tail -1
