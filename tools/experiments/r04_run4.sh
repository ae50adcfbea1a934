#!/bin/bash
# round 4, run 4: K order of the conv (channel-block size) under the ping-pong and the lock-step tile; DMA-only ablation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
for cb in 64 128 320; do UR_TEST_CBLOCK=$cb python tools/pp_ablate_time.py 9,32,49; done
for cb in 64 320; do UR_TEST_CBLOCK=$cb UR_LIB_PATH=$PWD/gpurun_ab/liburhip_ppabl6.so python tools/pp_ablate_time.py 49; done
} > gpurun_out/r04/pp_cblock.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/pp_cblock.txt
