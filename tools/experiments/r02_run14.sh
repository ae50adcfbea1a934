#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -m gpu -x -q > gpurun_out/r14_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r14_pytest.log
tail -4 gpurun_out/r14_pytest.log | cut -c1-300
python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-260
python tools/train_bench.py --steps 5 2>/dev/null | tail -1 | cut -c1-260
