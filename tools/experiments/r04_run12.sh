#!/bin/bash
# round 4, run 12: the dx-tap-sharing conv kernel (csrc/igemm_dxs.hip): parity, isolated timing, same-box A/B in the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "shared_dx_taps" 2>&1 | tail -15
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
{
for i in 1 2; do
echo "UR_DXS=0"; UR_DXS=0 $B | cut -c1-120
echo "UR_DXS=1"; UR_DXS=1 $B | cut -c1-120
done
echo "== isolated UR_DXS=0"; UR_DXS=0 python tools/pp_ablate_time.py 9,5
echo "== isolated UR_DXS=1"; UR_DXS=1 python tools/pp_ablate_time.py 9,5
} > gpurun_out/r04/dxs_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/dxs_ab.txt | sed -E 's/\{"metric".*"value": ([0-9.]+),.*/  \1 steps\/s/'
