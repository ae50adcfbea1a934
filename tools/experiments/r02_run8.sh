#!/bin/bash
# retune in the L2-cold ("scrubbed") regime with every tile family; bench base vs scrub-tuned table on the same box
mkdir -p gpurun_out
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r8_bench_base.json 2>/dev/null
cp uni_renderer_amd/igemm_tuning.json gpurun_out/igemm_tuning_scrub.json
timeout 3000 python tools/tune_igemm.py --scrub --tiles 1,2,3,4,5,6,7,8,9,10,11,13,14,15,22,23,24,31,32,33,34,36,37,38,40 --out gpurun_out/igemm_tuning_scrub.json --report gpurun_out/tune_report_scrub.json > gpurun_out/r8_tune.log 2>&1
tail -2 gpurun_out/r8_tune.log
UR_IGEMM_TUNING=gpurun_out/igemm_tuning_scrub.json python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r8_bench_scrub.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r8_bench_base2.json 2>/dev/null
for f in r8_bench_base r8_bench_scrub r8_bench_base2; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
