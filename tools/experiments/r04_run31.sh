#!/bin/bash
# round 4, run 31: in-situ tile tuning of the captured TRAINING step (forward + dgrad problems of ur_igemm; the weight gradients
# have their own table)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 3300 python tools/tune_in_situ.py --train --top 40 --cands 6 --out gpurun_out/r04/igemm_tuning_train.json > gpurun_out/r04/insitu_train.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/insitu_train.txt | grep -E "ACCEPT|in-situ"
