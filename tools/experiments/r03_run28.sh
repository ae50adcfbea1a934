#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
time python bench.py > gpurun_out/r03_bench_live_traffic.json 2> gpurun_out/r03_bench_live_traffic.err
tail -1 gpurun_out/r03_bench_live_traffic.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))"
tail -3 gpurun_out/r03_bench_live_traffic.err
