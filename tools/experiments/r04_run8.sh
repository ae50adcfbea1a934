#!/bin/bash
# round 4, run 8: full GPU suite after the ABI-8 changes, then a bench line of this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 3300 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/gpu_suite.txt 2>&1
tail -8 gpurun_out/r04/gpu_suite.txt
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-live-traffic > gpurun_out/r04/bench_mid.json 2> gpurun_out/r04/bench_mid.err; cut -c1-300 gpurun_out/r04/bench_mid.json
