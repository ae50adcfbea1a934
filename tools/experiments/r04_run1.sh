#!/bin/bash
# round 4, run 1: smoke at both configurations, the new training / DDP tests, a quick bench line of this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
( python __graft_entry__.py smoke ) > gpurun_out/r04/smoke_tiny.txt 2>&1
( UR_SMOKE_CONFIG=sd python __graft_entry__.py smoke ) > gpurun_out/r04/smoke_sd.txt 2>&1
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "fp16_amp or grad_scaler or torch_ddp or two_rank" -s > gpurun_out/r04/new_train_tests.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-live-traffic --no-cpu-baseline --shape-table gpurun_out/r04/shape_base.json > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err
tail -5 gpurun_out/r04/smoke_tiny.txt gpurun_out/r04/smoke_sd.txt; tail -15 gpurun_out/r04/new_train_tests.txt; cut -c1-400 gpurun_out/r04/bench_base.json
