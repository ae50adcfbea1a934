#!/bin/bash
for cfg in "X=1" "UR_TCHAIN=0" "UR_PRECISE_RESIDUAL=0" "UR_FOLD_SHORTCUT=0" "UR_GN_FUSED_MAX_ROWS=0" "UR_CONV_CBLOCK=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/poison_check.py --latent 64 --batch 4 --gb 12 2>&1 | grep "0x7B"
done
for b in 1 2 3; do
  echo "== batch $b"
  timeout 300 python tools/poison_check.py --latent 64 --batch $b --gb 12 2>&1 | grep "0x7B"
done
