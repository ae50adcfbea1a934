#!/bin/bash
# End-of-round collection of round 6 in three parts (each its own gpurun call: a slow counter pass must not cost the others;
# raw rocprofv3 output goes to /tmp, only summaries to gpurun_out/).  `python tools/install_profiles.py r06` afterwards copies
# the summaries into profiles/.      Usage: bash tools/r06_final.sh tests|profiles|lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
export TMPDIR=/tmp
case "$1" in
tests)
  timeout 1300 python -m pytest tests -q -m gpu -rP > $O/final_pytest.log 2>&1; tail -1 $O/final_pytest.log
  timeout 300 python __graft_entry__.py smoke > $O/final_smoke.txt 2>&1; tail -4 $O/final_smoke.txt
  ;;
profiles)
  bash tools/collect_profiles.sh r06 > $O/final_collect.log 2>&1; tail -3 $O/final_collect.log
  ;;
lines)
  for cfg in "cfg2 --direction render --batch 2 --latent 32 --dtype bf16" "cfg5 --batch 1 --latent 128"; do
    set -- $cfg; name=$1; shift
    timeout 400 python bench.py --no-cpu-baseline --no-loop "$@" > $O/final_bench_$name.json 2> $O/final_bench_$name.err
  done
  for b in 5 8 10 20; do
    timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --batch $b > $O/final_bench_b$b.json 2>/dev/null
  done
  timeout 600 python tools/loop_bench.py > $O/final_loop_bench.json 2>/dev/null
  timeout 300 python tools/hoist_bench.py --shape-table $O/final_hoist_shapes.json > $O/final_hoist_bench.json 2>/dev/null
  timeout 300 python tools/vae_bench.py > $O/final_vae_bench.json 2>/dev/null
  timeout 600 python tools/train_bench.py --graph > $O/final_train_graph.json 2>/dev/null
  timeout 600 python tools/train_bench.py > $O/final_train_eager.json 2>/dev/null
  timeout 600 python tools/train_bench.py --graph --force-collectives --comm-dtype bf16 --algorithm rs_ag > $O/final_train_graph_rccl_w1_captured.json 2>/dev/null
  timeout 600 python tools/train_bench.py --graph --force-collectives --comm-dtype bf16 --algorithm rs_ag --no-overlap > $O/final_train_graph_rccl_w1_serial.json 2>/dev/null
  (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ur_train_stats -o r06t --output-format csv -- python tools/train_bench.py --steps 3 --graph > $O/final_train_under_rocprof.json 2> /dev/null)
  cp $(find /tmp/ur_train_stats -name "*kernel_stats.csv" | head -1) $O/final_train_kernel_stats.csv; rm -rf /tmp/ur_train_stats
  # the hoisted per-step graph under rocprofv3 (kernel-trace only): which kernels a loop step is made of
  (timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ur_hoist_stats -o r06h --output-format csv -- python tools/hoist_bench.py --replays 40 > /dev/null 2>&1)
  cp $(find /tmp/ur_hoist_stats -name "*kernel_stats.csv" | head -1) $O/final_hoist_kernel_stats.csv; rm -rf /tmp/ur_hoist_stats
  ls $O/final_* | wc -l
  ;;
*) echo "usage: r06_final.sh tests|profiles|lines"; exit 2 ;;
esac
