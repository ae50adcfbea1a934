#!/bin/bash
# Row threshold of the 320-channel chain kernels (fused.TCHAIN_MIN_ROWS): cfg 2 (4096 rows per launch) and batch 2 at 64x64 (16384 grouped /
# 8192 hoisted) with the chain forced on (tchain_min_rows=0) and off (a huge threshold), one box.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for thr in 0 1000000; do
    echo "cfg2 tchain_min_rows=$thr: $(UR_EXPERIMENT=tchain_min_rows=$thr timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --direction render --batch 2 --latent 32 --dtype bf16 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    echo "b2 L64 grouped tchain_min_rows=$thr: $(UR_EXPERIMENT=tchain_min_rows=$thr timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --batch 2 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    echo "b2 L64 hoisted tchain_min_rows=$thr: $(UR_EXPERIMENT=tchain_min_rows=$thr B=2 timeout 300 python tools/hoist_bench.py --batch 2 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
