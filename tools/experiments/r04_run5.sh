#!/bin/bash
# round 4, run 5: schedule variants of the ping-pong kernel (setprio on / off, copies in the READ or in the MFMA block)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
python tools/pp_ablate_time.py 9,32,49,50
for v in 0 1 2 3; do UR_LIB_PATH=$PWD/gpurun_ab/liburhip_ppv$v.so python tools/pp_ablate_time.py 49,50,51,55; done
} > gpurun_out/r04/pp_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/pp_variants.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pingpong or (test_conv3x3 and (49 or 50 or 51 or 52 or 53 or 54 or 55)) or (linear_bias and (49 or 51 or 53 or 55)) or (geglu and (49 or 52 or 54)) or tail" 2>&1 | tail -5
