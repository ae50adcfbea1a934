#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift
  ms=$(env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$name $ms" | tee -a gpurun_out/knob_ab2.txt
}
: > gpurun_out/knob_ab2.txt
run base A=1
run exchange_late UR_EXCHANGE_EARLY=0
run ctx3_late UR_CTX3_EARLY=0
run vt_after_qk UR_VT_FIRST=0
run both_late UR_EXCHANGE_EARLY=0 UR_CTX3_EARLY=0
run base_end A=1
