#!/bin/bash
# end-of-round collection: full GPU suite (printing parity numbers), profiles (tag r02), other configs, loops, training
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -rP > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.log
grep -E "passed|failed|rc=" gpurun_out/final_pytest.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/collect_profiles.sh r02 > gpurun_out/final_collect.log 2>&1
python bench.py --batch 1 --latent 128 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_cfg5.json 2>/dev/null
python bench.py --batch 2 --latent 32 --dtype bf16 --direction render --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_cfg2.json 2>/dev/null
for b in 5 8 10 20; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_b$b.json 2>/dev/null; done
python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_graph.json 2>/dev/null
python tools/train_bench.py --steps 3 > gpurun_out/final_train_eager.json 2>/dev/null
python tools/loop_bench.py > gpurun_out/final_loop_bench.json 2>/dev/null
python tools/vae_bench.py > gpurun_out/final_vae_bench.json 2>/dev/null
python tools/attn_bwd_bench.py > /dev/null 2>&1; cp gpurun_out/attn_bwd_bench.json gpurun_out/final_attn_bwd_bench.json
python tools/train_bench.py --steps 3 --graph --torch-adamw > gpurun_out/final_train_graph_torch_adamw.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_under_rocprof.json 2>/dev/null; cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/final_train_kernel_stats.csv; rm -rf gpurun_out/prof_train)
tail -1 gpurun_out/prof_r02/bench_default.json | cut -c1-250
for f in cfg5 cfg2 b5 b8 b10 b20; do python -c "
import json
d=json.loads(open('gpurun_out/final_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
tail -1 gpurun_out/final_train_graph.json | cut -c1-200; tail -1 gpurun_out/final_loop_bench.json | cut -c1-300
