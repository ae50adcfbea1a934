#!/bin/bash
# round 4, run 16: ur_wgrad (LDS transpose reads) -- probe of the instruction, parity, training tests, same-box A/B of the
# graphed training step with and without it; forward step with the re-tuned table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
./tools/ubench/tr_probe > gpurun_out/r04/tr_probe.txt 2>&1; tail -1 gpurun_out/r04/tr_probe.txt
timeout 1500 python -m pytest tests/test_wgrad_gpu.py -x -q 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for f in 1 0; do
    echo "UR_WGRAD=$f"; UR_WGRAD=$f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-260
  done
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
