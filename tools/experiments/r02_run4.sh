#!/bin/bash
# round-2 GPU run 4: new multi-rank test, full profile collection (tag r02), other configs, training step
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py -m gpu -x -q -rP -k "two_rank or module_surface" > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest.log
grep -E "passed|failed|rc=|two_rank" gpurun_out/r4_pytest.log | tail -4 | cut -c1-600
bash tools/collect_profiles.sh r02 > gpurun_out/r4_collect.log 2>&1
tail -3 gpurun_out/r4_collect.log | cut -c1-300
python bench.py --batch 1 --latent 128 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r4_bench_cfg5.json 2>/dev/null
python bench.py --batch 2 --latent 32 --dtype bf16 --direction render --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r4_bench_cfg2.json 2>/dev/null
python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r4_bench_b8.json 2>/dev/null
python tools/train_bench.py --steps 3 --graph > gpurun_out/r4_train_graph.json 2>/dev/null
python tools/train_bench.py --steps 3 > gpurun_out/r4_train_eager.json 2>/dev/null
python tools/loop_bench.py > gpurun_out/r4_loop_bench.json 2>/dev/null
for f in r4_bench_cfg5 r4_bench_cfg2 r4_bench_b8; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
tail -1 gpurun_out/r4_train_graph.json | cut -c1-300; tail -1 gpurun_out/r4_train_eager.json | cut -c1-300; tail -2 gpurun_out/r4_loop_bench.json | cut -c1-300
