#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift
  ms=$(env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$name $ms" | tee -a gpurun_out/knob_ab3.txt
}
: > gpurun_out/knob_ab3.txt
run norm_xcd_on UR_NORM_XCD=1
run norm_xcd_off UR_NORM_XCD=0
run norm_xcd_on2 UR_NORM_XCD=1
run norm_xcd_off2 UR_NORM_XCD=0
python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py -m gpu -x -q -k "groupnorm or layernorm or gn or ln or grouped" 2>&1 | tail -2
