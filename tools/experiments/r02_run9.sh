#!/bin/bash
# in-situ A/B: the base table vs the same table with every 128x320 conv entry switched to its 4-loader-wave build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in base l4; do
  if [ $tag = l4 ]; then export UR_IGEMM_TUNING=$R/tools/data/igemm_tuning_9to32.json; else unset UR_IGEMM_TUNING; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r9_${tag}_$rep.json 2>/dev/null
  python -c "
import json
print('$tag $rep', json.loads(open('gpurun_out/r9_${tag}_$rep.json').read().strip().splitlines()[-1])['ms_per_step'])"
done; done
for tag in base l4; do
  if [ $tag = l4 ]; then export UR_IGEMM_TUNING=$R/tools/data/igemm_tuning_9to32.json; else unset UR_IGEMM_TUNING; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab9_$tag -o ab --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  cp $(find $R/gpurun_out/ab9_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r9_kernel_stats_$tag.csv
  rm -rf $R/gpurun_out/ab9_$tag
done
