#!/bin/bash
# round-2 GPU run 2: M32 (32x32x16 MFMA) igemm tiles: correctness, then a full retune with them in the candidate list, bench
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest.log
tail -5 gpurun_out/r2_pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_base.json 2> gpurun_out/r2_bench_base.err
UR_SIDE_STREAM=2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_side2.json 2> gpurun_out/r2_bench_side2.err
cp uni_renderer_amd/igemm_tuning.json gpurun_out/igemm_tuning_m32.json
timeout 2400 python tools/tune_igemm.py --out gpurun_out/igemm_tuning_m32.json --report gpurun_out/tune_report_m32.json > gpurun_out/r2_tune.log 2>&1
UR_IGEMM_TUNING=gpurun_out/igemm_tuning_m32.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-table > gpurun_out/r2_bench_m32.json 2> gpurun_out/r2_bench_m32.err
for f in r2_bench_base r2_bench_side2 r2_bench_m32; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
