#!/bin/bash
# round 4, run 15: in-situ tuning pass over the 30 heaviest problems of the final tree (new merged-projection shapes included)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 4500 python tools/tune_in_situ.py --broad --tiles 9,10,1,5,7,2,3,11,8,32 --top 30 --replays 60 --out gpurun_out/r04/igemm_tuning_insitu4.json > gpurun_out/r04/insitu4.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/insitu4.txt | grep -E "ACCEPT|in-situ"
