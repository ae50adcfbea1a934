#!/bin/bash
# round-2 GPU run 1: new tests, side-stream A/B, full retune (sk up to 16), bench with the new table
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py tests/test_fused_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1_pytest.log
UR_SIDE_STREAM=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r1_bench_side0.json 2> gpurun_out/r1_bench_side0.err
UR_SIDE_STREAM=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r1_bench_side1.json 2> gpurun_out/r1_bench_side1.err
cp uni_renderer_amd/igemm_tuning.json gpurun_out/igemm_tuning_r02.json
timeout 1500 python tools/tune_igemm.py --out gpurun_out/igemm_tuning_r02.json --report gpurun_out/tune_report_r02.json > gpurun_out/r1_tune.log 2>&1
UR_IGEMM_TUNING=gpurun_out/igemm_tuning_r02.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r1_bench_retuned.json 2> gpurun_out/r1_bench_retuned.err
tail -3 gpurun_out/r1_pytest.log
for f in gpurun_out/r1_bench_side0.json gpurun_out/r1_bench_side1.json gpurun_out/r1_bench_retuned.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
