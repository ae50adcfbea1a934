#!/bin/bash
# round 4, run 23: ur_wgrad with 2-stage rings and <= 128 VGPRs (4 waves per SIMD): parity, per-problem sweep over tiles / depths / slices
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -2
timeout 1500 python tools/wgrad_bench.py --sweep --tiles 1,2,3,4,5,6,9,11,12,13 --splits 1,2,4,8,16,32 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/wgrad_bench4.txt
tail -1 gpurun_out/r04/wgrad_bench4.txt
