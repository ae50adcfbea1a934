#!/bin/bash
# round-2 GPU run 3: full GPU suite with printed parity numbers, VAE timing, tuning rows for the repeat batches (5/10/20)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -rP > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest.log
grep -E "passed|failed|rc=" gpurun_out/r3_pytest.log | tail -3
python tools/vae_bench.py --batch 4 > gpurun_out/r3_vae_bench.json 2>&1; tail -1 gpurun_out/r3_vae_bench.json
cp uni_renderer_amd/igemm_tuning.json gpurun_out/igemm_tuning_r3.json
timeout 2400 python tools/tune_igemm.py --batch 5 --also "10,64;20,64" --only-missing --tiles 1,2,3,4,5,6,7,8,9,10,11 --out gpurun_out/igemm_tuning_r3.json --report gpurun_out/tune_report_r3.json > gpurun_out/r3_tune.log 2>&1
tail -2 gpurun_out/r3_tune.log
for b in 5 10 20; do UR_IGEMM_TUNING=gpurun_out/igemm_tuning_r3.json python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_b$b.json 2>/dev/null; python -c "
import json
d=json.loads(open('gpurun_out/r3_bench_b$b.json').read().strip().splitlines()[-1]); print('batch $b', d['ms_per_step'])"; done
