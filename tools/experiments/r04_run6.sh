#!/bin/bash
# round 4, run 6: in-situ tuning pass with the ping-pong tiles as candidates (25 heaviest problems)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 2400 python tools/tune_in_situ.py --broad --tiles 50,55,53,51 --top 25 --replays 60 --out gpurun_out/r04/igemm_tuning_pp.json > gpurun_out/r04/insitu_pp.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/insitu_pp.txt | grep -E "ACCEPT|in-situ"
