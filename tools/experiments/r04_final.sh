#!/bin/bash
# end-of-round collection (round 4): full GPU suite (printing parity numbers), smoke, profiles (tag r04), the other
# single-GPU configurations WITH roofline + live traffic, loops, full-loop parity, training, determinism
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/final_*
python -m pytest tests -m gpu -x -q -rP > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.log
grep -E "passed|failed|rc=" gpurun_out/final_pytest.log | tail -3
python __graft_entry__.py smoke > gpurun_out/final_smoke.txt 2>&1; tail -4 gpurun_out/final_smoke.txt
bash tools/collect_profiles.sh r04 > gpurun_out/final_collect.log 2>&1
python bench.py --batch 1 --latent 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_cfg5.json 2>/dev/null
python bench.py --batch 2 --latent 32 --dtype bf16 --direction render --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_cfg2.json 2>/dev/null
for b in 5 8 10 20; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/final_bench_b$b.json 2>/dev/null; done
python tools/train_bench.py --steps 5 --graph > gpurun_out/final_train_graph.json 2>/dev/null
UR_WGRAD=0 python tools/train_bench.py --steps 5 --graph > gpurun_out/final_train_graph_transposed_path.json 2>/dev/null
python tools/wgrad_bench.py > gpurun_out/final_wgrad_bench.txt 2>&1
python tools/train_bench.py --steps 3 > gpurun_out/final_train_eager.json 2>/dev/null
python tools/loop_bench.py > gpurun_out/final_loop_bench.json 2>/dev/null
python tools/vae_bench.py > gpurun_out/final_vae_bench.json 2>/dev/null
python tools/loop_parity.py --out gpurun_out/final_loop_parity.json > gpurun_out/final_loop_parity.log 2>&1; tail -1 gpurun_out/final_loop_parity.log | cut -c1-400
{ echo "== step, latent 64 batch 4"; python tools/determinism_check.py --n 200 2>&1 | tail -1; echo "== step, latent 128 batch 1"; python tools/determinism_check.py --latent 128 --batch 1 --n 60 2>&1 | tail -1; } > gpurun_out/final_determinism.txt 2>&1
(export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1; python tools/gap_analysis.py "gpurun_out/trace/*/*kernel_trace.csv" --steps 10 > gpurun_out/final_gap_analysis.txt 2>&1; rm -rf gpurun_out/trace)
(export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o tr --output-format csv -- python tools/train_bench.py --steps 3 --graph > gpurun_out/final_train_under_rocprof.json 2>/dev/null; cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) gpurun_out/final_train_kernel_stats.csv; rm -rf gpurun_out/prof_train)
tail -1 gpurun_out/prof_r04/bench_default.json | cut -c1-250
for f in cfg5 cfg2 b5 b8 b10 b20; do python -c "
import json
d=json.loads(open('gpurun_out/final_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'))"; done
tail -1 gpurun_out/final_train_graph.json | cut -c1-200; tail -1 gpurun_out/final_loop_bench.json | cut -c1-300; head -3 gpurun_out/final_gap_analysis.txt
