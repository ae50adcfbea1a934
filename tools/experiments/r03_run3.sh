#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench1.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'sum_kernel_ms', d['config'].get('sum_kernel_ms_eager_step'))
for r in d['kernel_classes']: print(r['kernel'], r['calls'], r['ms'], r.get('tflops', r.get('gbs')))
PY
tail -3 gpurun_out/r03_bench1.err
UR_TCHAIN=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('UR_TCHAIN=0 ms_per_step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
timeout 1500 python -m pytest tests/test_configs_gpu.py -x -q -rP -k "cfg3 or cfg2" 2>&1 | grep -E "rel|passed|failed|Error|assert" | tail -20
