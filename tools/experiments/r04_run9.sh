#!/bin/bash
# round 4, run 9: fused split-K reduce + GroupNorm, merged prompt K | V^T projection -- op parity, tuning of the new shapes,
# same-box A/B of each switch, step parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_groupnorm or qkv_projection" 2>&1 | tail -4
timeout 900 python tools/tune_igemm.py --only-missing --tiles 1,2,3,5,7,8,9,10,11,24,13,32 > gpurun_out/r04/tune_ctxkv.txt 2>&1
grep -E "M=|wrote" gpurun_out/r04/tune_ctxkv.txt | tail -6
cp uni_renderer_amd/igemm_tuning.json gpurun_out/r04/igemm_tuning_ctxkv.json
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2; do
echo "base (both off)";  UR_SPLITK_GN=0 UR_CTXKV_ONE_LAUNCH=0 $B | cut -c1-120
echo "splitk_gn only";   UR_SPLITK_GN=1 UR_CTXKV_ONE_LAUNCH=0 $B | cut -c1-120
echo "ctxkv only";       UR_SPLITK_GN=0 UR_CTXKV_ONE_LAUNCH=1 $B | cut -c1-120
echo "both (default)";   $B | cut -c1-120
done
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5
