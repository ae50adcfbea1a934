#!/bin/bash
# Bucketed training step, round-6 gradient protocol (direct writes into the buckets) against round 5's (zero + accumulate), one box.
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out
for rep in 1 2; do
  timeout 500 python tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype bf16 --algorithm rs_ag 2>/dev/null | head -1 > $O/r06_train_rccl_direct_$rep.json
  timeout 500 python tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype bf16 --algorithm rs_ag --accumulate-into-buckets 2>/dev/null | head -1 > $O/r06_train_rccl_accum_$rep.json
done
timeout 500 python tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype bf16 --algorithm rs_ag --no-overlap 2>/dev/null | head -1 > $O/r06_train_rccl_direct_serial.json
timeout 500 python tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype fp32 --algorithm all_reduce 2>/dev/null | head -1 > $O/r06_train_rccl_direct_fp32.json
timeout 500 python tools/train_bench.py --graph --steps 3 2>/dev/null | head -1 > $O/r06_train_graph.json
for f in $O/r06_train_rccl_direct_1.json $O/r06_train_rccl_accum_1.json $O/r06_train_rccl_direct_2.json $O/r06_train_rccl_accum_2.json $O/r06_train_rccl_direct_serial.json $O/r06_train_rccl_direct_fp32.json $O/r06_train_graph.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['ms_per_step'], d['loss'], d['grad_norm'], d.get('phase_ms'))"; done
