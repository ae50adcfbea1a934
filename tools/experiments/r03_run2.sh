#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tchain_gpu.py -x -q -rP 2>&1 | grep -E "tchain_|passed|failed|Error|error" | tail -40
timeout 300 python tools/tchain_bench.py 2>&1 | tail -1
