#!/bin/bash
# round 4, run 2: parity of the ping-pong tiles, then their isolated A/B against the shipped table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pingpong or (test_conv3x3 and (49 or 50 or 51 or 52 or 53 or 54 or 55)) or (linear_bias and (49 or 51 or 53 or 55)) or (geglu and (49 or 52 or 54)) or tail" > gpurun_out/r04/pp_tests.txt 2>&1
tail -15 gpurun_out/r04/pp_tests.txt
timeout 1500 python tools/pp_ab.py --min-us 20 > gpurun_out/r04/pp_ab.txt 2>&1
tail -70 gpurun_out/r04/pp_ab.txt
