#!/bin/bash
# in-step A/B of the weight-streaming conv policy (same box, alternating)
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "UR_WSCONV=0" "UR_WSCONV=1 UR_WSCONV_MIN_K=5000" "UR_WSCONV=1 UR_WSCONV_MIN_K=8000" "UR_WSCONV=1 UR_WSCONV_MIN_K=0"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
done
timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -k "cfg3 or cfg2" 2>&1 | tail -3
