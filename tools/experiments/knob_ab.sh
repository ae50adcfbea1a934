#!/bin/bash
# in-situ A/B of the host-side knobs (env vars) on one box: each line = one bench.py run of 100 replays
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  ms=$(env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$name $ms" | tee -a gpurun_out/knob_ab.txt
}
: > gpurun_out/knob_ab.txt
run base A=1
run base_again A=1
run gn_fused_1024 UR_GN_FUSED_MAX_ROWS=1024
run gn_fused_64 UR_GN_FUSED_MAX_ROWS=64
run gn_fused_0 UR_GN_FUSED_MAX_ROWS=0
run cblock_0 UR_CONV_CBLOCK=0
run cblock_640 UR_CONV_CBLOCK=640
run fold_shortcut_0 UR_FOLD_SHORTCUT=0
run gn_stat_32k UR_GN_STAT_KB=32
run gn_stat_128k UR_GN_STAT_KB=128
run gn_apply_10k UR_GN_APPLY_KB=10
run gn_apply_40k UR_GN_APPLY_KB=40
run gn_apply_max256 UR_GN_APPLY_MAX=256
run gn_apply_max64 UR_GN_APPLY_MAX=64
run plain_residual UR_PRECISE_RESIDUAL=0
run base_end A=1
