#!/bin/bash
# round 4, run 20: ur_wgrad, tile order within a slice by the shorter dimension: per-problem sweep again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -2
timeout 1500 python tools/wgrad_bench.py --sweep 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/wgrad_bench3.txt
cut -c1-460 gpurun_out/r04/wgrad_bench3.txt
