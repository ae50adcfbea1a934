#!/bin/bash
# Per-kernel totals of the captured cfg-2 step (rendering direction, 256x256, bs 2, bf16) under rocprofv3 --kernel-trace --stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_cfg2
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_cfg2 -o s --output-format csv -- python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --direction render --batch 2 --latent 32 --dtype bf16 > $R/gpurun_out/r06_cfg2_under_rocprof.json 2>/dev/null)
cp $(find /tmp/rp_cfg2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_cfg2_kernel_stats.csv
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$R/gpurun_out/r06_cfg2_kernel_stats.csv')) if 'ur' in r['Name'][:12]]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('our kernels total ms', tot/1e6)
for r in rows[:32]:
    print('%-86s calls %5s total %7.2f ms avg %6.1f us  %4.1f%%'%(r['Name'][:86], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
