#!/bin/bash
# round 4, run 32: 32-deep-chunk tiles of ur_igemm (four workgroups per CU): parity over every tile test, isolated A/B against the
# table on the step's problems
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -k "56 or 57 or 58 or 59 or 60 or 61" 2>&1 | tail -3
timeout 1500 python tools/pp_ab.py --tiles 56,57,58,59,60,61 --max-wgs 4096 --out gpurun_out/r04/k32_ab.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/k32_ab.txt
cut -c1-330 gpurun_out/r04/k32_ab.txt
