#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/r03_bench_live_traffic.json 2> gpurun_out/r03_bench_live_traffic.err ) 2>&1 | grep real
tail -1 gpurun_out/r03_bench_live_traffic.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60]); print(d['cpu_baseline']['value'])"
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
