#!/bin/bash
# round 4, run 3: ablation builds of the ping-pong kernel (what bounds it?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
python tools/pp_ablate_time.py 9,49,50,55
for a in 1 2 4 3 5 6 8; do UR_LIB_PATH=$PWD/gpurun_ab/liburhip_ppabl$a.so python tools/pp_ablate_time.py 49,55; done
} > gpurun_out/r04/pp_ablate.txt 2>&1
cat gpurun_out/r04/pp_ablate.txt
