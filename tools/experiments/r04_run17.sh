#!/bin/bash
# round 4, run 17: ur_wgrad per problem against the transposed-operand path, tiles x slice counts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python tools/wgrad_bench.py --sweep 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/wgrad_bench.txt
cat gpurun_out/r04/wgrad_bench.txt
