#!/bin/bash
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03_full_gpu_suite_2.log
cat gpurun_out/r03_full_gpu_suite_2.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
