#!/bin/bash
# round 4, run 22: ur_wgrad ring depth (library built with make WGRAD_ABL=1: tiles 7..11 = the product tiles at other depths)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -2
timeout 1500 python tools/wgrad_bench.py --sweep --tiles 8,1,7,9,4,11,5,10 --splits 1,4,16 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/wgrad_ring.txt
cut -c1-520 gpurun_out/r04/wgrad_ring.txt
