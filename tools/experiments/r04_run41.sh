#!/bin/bash
# round 4, run 41: pack kernel back to the direct form, two-level fold in the one-launch GroupNorm backward: parity + step time
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "pack_unpack or groupnorm" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -k "graph or oracle_autograd or sd_size" 2>&1 | tail -2
for i in 1 2 3; do python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > /dev/null 2>&1
mkdir -p gpurun_out/r04; cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/r04/train_kernel_stats_41.csv; rm -rf gpurun_out/prof_train
grep -E "gn_bwd_fused|pack_conv" gpurun_out/r04/train_kernel_stats_41.csv | cut -c1-160
