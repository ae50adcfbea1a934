#!/bin/bash
# wsconv bring-up: parity tests, new igemm tiles, isolated timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wsconv_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r03_wsconv_test.log
cat gpurun_out/r03_wsconv_test.log
timeout 600 python tools/wsconv_bench.py --iters 20 > gpurun_out/r03_wsconv_bench.log 2>&1
cat gpurun_out/r03_wsconv_bench.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r03_ops_test.log
cat gpurun_out/r03_ops_test.log
