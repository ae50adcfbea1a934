#!/bin/bash
# round 4, run 24: tune ur_wgrad on the problems of the cfg 4 training step, then the same-box A/B of the graphed step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python tools/tune_wgrad.py --out gpurun_out/r04/wgrad_tuning.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/tune_wgrad.txt
tail -3 gpurun_out/r04/tune_wgrad.txt
cp gpurun_out/r04/wgrad_tuning.json uni_renderer_amd/wgrad_tuning.json
timeout 900 python -m pytest tests/test_wgrad_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
  for f in 1 0; do
    echo "UR_WGRAD=$f"; UR_WGRAD=$f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
  done
done
echo "UR_WGRAD=1 UR_WGRAD_TABLE=0"; UR_WGRAD_TABLE=0 python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
