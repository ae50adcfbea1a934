#!/bin/bash
# refresh the training part of the collection with the final code
mkdir -p gpurun_out
python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_graph.json 2>/dev/null
python tools/train_bench.py --steps 3 > gpurun_out/final_train_eager.json 2>/dev/null
python tools/train_bench.py --steps 3 --graph --torch-adamw > gpurun_out/final_train_graph_torch_adamw.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_under_rocprof.json 2>/dev/null; cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/final_train_kernel_stats.csv; rm -rf gpurun_out/prof_train)
python tools/train_determinism.py --steps 3 2>&1 | tail -2 > gpurun_out/final_train_determinism.txt
for f in graph eager graph_torch_adamw; do tail -1 gpurun_out/final_train_$f.json | cut -c1-160; done
head -8 gpurun_out/final_train_kernel_stats.csv | cut -c1-150
