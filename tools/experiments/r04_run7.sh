#!/bin/bash
# round 4, run 7: merged q|k|v projection -- op parity, step parity, tuning of the three new shapes, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "qkv_projection" 2>&1 | tail -4
timeout 1200 python tools/tune_igemm.py --only-missing --tiles 1,2,3,5,7,8,9,10,11,24,13,32 > gpurun_out/r04/tune_qkv.txt 2>&1
grep -E "M=|wrote" gpurun_out/r04/tune_qkv.txt | tail -12
cp uni_renderer_amd/igemm_tuning.json gpurun_out/r04/igemm_tuning_qkv.json
for i in 1 2; do
UR_QKV_ONE_LAUNCH=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline | cut -c1-140
UR_QKV_ONE_LAUNCH=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline | cut -c1-140
done
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_fused_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5
