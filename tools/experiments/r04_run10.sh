#!/bin/bash
# round 4, run 10: XCD placement of the one-launch GroupNorm and of the fused split-K reduce + GroupNorm -- parity, A/B;
# tuning rows for the merged-projection shapes of the other configurations (cfg 2, cfg 5, batches 1 / 5 / 8 / 10 / 20)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_groupnorm or groupnorm or gn" 2>&1 | tail -3
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2; do
echo "gnf_xcd=0";              UR_GNF_XCD=0 $B | cut -c1-120
echo "gnf_xcd=1 (default)";    $B | cut -c1-120
echo "gnf_xcd=1 splitk_gn=1";  UR_SPLITK_GN=1 $B | cut -c1-120
done
echo "== gn_bench UR_GNF_XCD=0"; UR_GNF_XCD=0 python tools/gn_bench.py 2>&1 | grep -v amdgpu | tail -10
echo "== gn_bench UR_GNF_XCD=1"; python tools/gn_bench.py 2>&1 | grep -v amdgpu | tail -10
timeout 2400 python tools/tune_igemm.py --only-missing --batch 2 --latent 32 --also "1,128;1,64;5,64;8,64;10,64;20,64" --tiles 1,2,3,5,7,8,9,10,11,24,13,32 > gpurun_out/r04/tune_other_cfgs.txt 2>&1
grep -E "distinct|wrote" gpurun_out/r04/tune_other_cfgs.txt | tail -10
cp uni_renderer_amd/igemm_tuning.json gpurun_out/r04/igemm_tuning_other.json
