#!/bin/bash
# round 4, run 27: 3x3 conv weight gradients deferred + grouped as well (PackConvWeights node): parity, tuning, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -3
timeout 1500 python tools/tune_wgrad.py --out gpurun_out/r04/wgrad_tuning3.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/tune_wgrad3.txt
tail -2 gpurun_out/r04/tune_wgrad3.txt
cp gpurun_out/r04/wgrad_tuning3.json uni_renderer_amd/wgrad_tuning.json
for i in 1 2; do
  for f in "UR_WGRAD_DEFER=1" "UR_WGRAD_DEFER=0" "UR_WGRAD=0"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-260
  done
done
