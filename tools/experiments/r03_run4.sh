#!/bin/bash
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_pipeline_gpu.py tests/test_configs_gpu.py::test_cfg3_five_step_ddim_loop_at_sd_size_batch_4 tests/test_train_gpu.py -x -q -rP -k "not two_rank" > gpurun_out/r03_run4.log 2>&1
grep -E "^\{|passed|failed|Error|assert " gpurun_out/r03_run4.log | cut -c1-400 | tail -40
