#!/bin/bash
# round 4, run 19: ur_wgrad with 256-wide tiles (8 / 16 waves): parity, per-problem sweep, ablation of the 256 x 256 tile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -5
timeout 1500 python tools/wgrad_bench.py --sweep 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/wgrad_bench2.txt
cat gpurun_out/r04/wgrad_bench2.txt
{ for t in 4 5; do echo "== tile $t"; python tools/wgrad_ablate.py --tile $t 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r04/wgrad_ablate2.txt
cat gpurun_out/r04/wgrad_ablate2.txt
