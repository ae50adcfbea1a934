#!/bin/bash
# which of the changes since the first in-situ pass carries the gain: table (pre/post) x launch-order / norm knobs
mkdir -p gpurun_out; : > gpurun_out/ab_knobs.txt
run() {  # tag, env...
  tag=$1; shift
  ms=$(env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$tag $ms" | tee -a gpurun_out/ab_knobs.txt
}
PRE=UR_IGEMM_TUNING=$GRAFT_REPO_ROOT/tools/data/igemm_tuning_pre_insitu.json
OLD="UR_NORM_XCD=0 UR_EXCHANGE_EARLY=0 UR_CTX3_EARLY=0 UR_VT_FIRST=0"
for i in 1 2; do
run post_default X=1
run pre_default $PRE
run pre_oldorder $PRE $OLD
run post_oldorder $OLD
run pre_noxcd $PRE UR_NORM_XCD=0
run pre_noearly $PRE UR_EXCHANGE_EARLY=0 UR_CTX3_EARLY=0
run pre_novt $PRE UR_VT_FIRST=0
done
