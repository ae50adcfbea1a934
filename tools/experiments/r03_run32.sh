#!/bin/bash
for rep in 1 2; do
for cfg in "UR_SIDE_STREAM=0" "UR_SIDE_STREAM=1" "UR_SIDE_STREAM=2" "UR_SIDE_STREAM=3"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
done
