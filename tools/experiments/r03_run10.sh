#!/bin/bash
# wsconv experiments (debug build): what bounds the stage time -- operand traffic, weight traffic, or neither
for a in 0 101 102 103; do
  echo "== UR_WS_DEBUG_ACT=$a"
  UR_WS_DEBUG_ACT=$a timeout 300 python tools/wsconv_bench.py --iters 20 --sk 1 --only 0,1,3 2>&1 | grep -v amdgpu.ids
done
