#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/trace_bench.log 2>&1
tail -1 gpurun_out/trace_bench.log | cut -c1-300
ls gpurun_out/trace/*/ | head
python tools/gap_analysis.py "gpurun_out/trace/*/*kernel_trace.csv" --steps 10 | tee gpurun_out/r03_gap_analysis.txt
rm -rf gpurun_out/trace
