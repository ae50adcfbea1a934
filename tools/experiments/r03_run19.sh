#!/bin/bash
for cfg in "X=1" "UR_TCHAIN=0"; do
  echo "== $cfg"
  env $cfg timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -k "cfg5_relighting" -s 2>&1 | grep -E '^\{"cfg"|passed|failed' | cut -c1-700
done
