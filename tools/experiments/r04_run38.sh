#!/bin/bash
# round 4, run 38: d = 40 attention backward reading the head rows in place (no per-head copies) now that no transposes are needed
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  for f in "UR_FLASH_DIRECT_MIN_D=64" "UR_FLASH_DIRECT_MIN_D=40"; do
    echo "$f"; env $f python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
  done
done
