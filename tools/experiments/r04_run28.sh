#!/bin/bash
# round 4, run 28: kernel statistics of the graphed training step with ur_wgrad
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > gpurun_out/r04/train_under_rocprof.json 2>/dev/null
cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/r04/train_kernel_stats_wgrad.csv
rm -rf gpurun_out/prof_train
head -5 gpurun_out/r04/train_kernel_stats_wgrad.csv | cut -c1-200
