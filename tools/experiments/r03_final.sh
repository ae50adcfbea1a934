#!/bin/bash
# end-of-round collection: full GPU suite (printing parity numbers), profiles (tag r03), other configs, loops, training
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -rP > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.log
grep -E "passed|failed|rc=" gpurun_out/final_pytest.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/collect_profiles.sh r03 > gpurun_out/final_collect.log 2>&1
python bench.py --batch 1 --latent 128 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_cfg5.json 2>/dev/null
python bench.py --batch 2 --latent 32 --dtype bf16 --direction render --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_cfg2.json 2>/dev/null
for b in 5 8 10 20; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_b$b.json 2>/dev/null; done
python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_graph.json 2>/dev/null
python tools/train_bench.py --steps 3 > gpurun_out/final_train_eager.json 2>/dev/null
python tools/loop_bench.py > gpurun_out/final_loop_bench.json 2>/dev/null
python tools/vae_bench.py > gpurun_out/final_vae_bench.json 2>/dev/null
python tools/tchain_bench.py > gpurun_out/final_tchain_bench.json 2>/dev/null
{ echo "== step, latent 64 batch 4"; python tools/determinism_check.py --n 200 2>&1 | tail -1; echo "== step, latent 128 batch 1"; python tools/determinism_check.py --latent 128 --batch 1 --n 60 2>&1 | tail -1; echo "== chains"; python tools/tchain_determinism.py --iters 1000 2>&1 | tail -3; } > gpurun_out/final_determinism.txt 2>&1
python tools/wsconv_bench.py --iters 20 > gpurun_out/final_wsconv_bench.txt 2>&1
bash tools/pmc_conv_sq.sh gpurun_out/final_pmc_conv_sq.json > /dev/null 2>&1
./tools/ubench/mfma_rate > gpurun_out/final_mfma_rate.txt 2>&1
bash tools/experiments/r03_run6.sh > gpurun_out/final_ring_depth_ab.txt 2>&1
python tools/train_bench.py --steps 3 --graph --torch-adamw > gpurun_out/final_train_graph_torch_adamw.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_under_rocprof.json 2>/dev/null; cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/final_train_kernel_stats.csv; rm -rf gpurun_out/prof_train)
tail -1 gpurun_out/prof_r03/bench_default.json | cut -c1-250
for f in cfg5 cfg2 b5 b8 b10 b20; do python -c "
import json
d=json.loads(open('gpurun_out/final_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
tail -1 gpurun_out/final_train_graph.json | cut -c1-200; tail -1 gpurun_out/final_loop_bench.json | cut -c1-300
