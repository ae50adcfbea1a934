#!/bin/bash
# round-2 GPU run 5: wave-specialised igemm tiles (dedicated loader waves): correctness, retune with them, bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
tail -4 gpurun_out/r5_pytest.log | cut -c1-300
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_base.json 2> gpurun_out/r5_bench_base.err
cp uni_renderer_amd/igemm_tuning.json gpurun_out/igemm_tuning_ws.json
timeout 2400 python tools/tune_igemm.py --tiles 1,2,3,4,5,7,8,9,10,11,31,32,33,34,35,36,37,38,40 --out gpurun_out/igemm_tuning_ws.json --report gpurun_out/tune_report_ws.json > gpurun_out/r5_tune.log 2>&1
tail -2 gpurun_out/r5_tune.log
UR_IGEMM_TUNING=gpurun_out/igemm_tuning_ws.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-table > gpurun_out/r5_bench_ws.json 2> gpurun_out/r5_bench_ws.err
for f in r5_bench_base r5_bench_ws; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
