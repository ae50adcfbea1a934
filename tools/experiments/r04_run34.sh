#!/bin/bash
# round 4, run 34: in-situ tuning of the cfg 5 step (B = 1, 128 x 128 latent) and of the B = 8 step (cfg 3 shapes at twice the batch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python tools/tune_in_situ.py --batch 1 --latent 128 --broad --tiles 9,10,1,5,7,2,3,11,8,32 --top 24 --replays 40 --out gpurun_out/r04/igemm_tuning_cfg5.json > gpurun_out/r04/insitu_cfg5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/insitu_cfg5.txt | grep -E "ACCEPT|in-situ"
cp gpurun_out/r04/igemm_tuning_cfg5.json uni_renderer_amd/igemm_tuning.json
timeout 1500 python tools/tune_in_situ.py --batch 8 --latent 64 --broad --tiles 9,10,1,5,7,2,3,11,8,32 --top 24 --replays 30 --out gpurun_out/r04/igemm_tuning_b8.json > gpurun_out/r04/insitu_b8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/insitu_b8.txt | grep -E "ACCEPT|in-situ"
