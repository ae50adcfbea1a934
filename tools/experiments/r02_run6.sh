#!/bin/bash
# A/B in situ under rocprofv3: base table vs wave-specialised table (same box)
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for tag in base ws; do
  if [ $tag = ws ]; then export UR_IGEMM_TUNING=$R/tools/data/igemm_tuning_ws.json; else unset UR_IGEMM_TUNING; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$tag -o ab --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r6_bench_$tag.json 2> gpurun_out/r6_$tag.err
  cp $(find $R/gpurun_out/ab_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r6_kernel_stats_$tag.csv
  rm -rf $R/gpurun_out/ab_$tag
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r6_bench_plain_$tag.json 2>/dev/null
  python -c "
import json
print('$tag', json.loads(open('gpurun_out/r6_bench_plain_$tag.json').read().strip().splitlines()[-1])['ms_per_step'])"
done
