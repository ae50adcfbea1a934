#!/bin/bash
for cfg in "X=1" "UR_TCHAIN=0"; do
  echo "== $cfg"
  env $cfg timeout 600 python tools/determinism_check.py --n 80 2>&1 | tail -6
done
