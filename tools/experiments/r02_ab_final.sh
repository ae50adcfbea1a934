#!/bin/bash
# same-box alternation: the table before the in-situ passes vs the shipped table
mkdir -p gpurun_out; : > gpurun_out/ab_final.txt
for i in 1 2 3; do
for tag in pre post; do
  if [ $tag = pre ]; then export UR_IGEMM_TUNING=$GRAFT_REPO_ROOT/tools/data/igemm_tuning_pre_insitu.json; else unset UR_IGEMM_TUNING; fi
  ms=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$tag $i $ms" | tee -a gpurun_out/ab_final.txt
done; done
