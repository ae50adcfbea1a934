#!/bin/bash
# round 4, run 14: padding rows read from a 1-MiB zero region (one line per workgroup and wave) instead of one hot 128-byte
# line; the dx-tap-sharing kernel with its permanent halos zeroed once in LDS -- parity, then a 2 x 2 A/B in the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "shared_dx_taps" 2>&1 | tail -4
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
{
for i in 1 2; do
echo "dxs=0 zero=4096"; UR_DXS=0 UR_ZERO_PAGE_BYTES=4096 $B | cut -c1-120
echo "dxs=0 zero=1M";   UR_DXS=0 $B | cut -c1-120
echo "dxs=1 zero=4096"; UR_DXS=1 UR_ZERO_PAGE_BYTES=4096 $B | cut -c1-120
echo "dxs=1 zero=1M";   UR_DXS=1 $B | cut -c1-120
done
echo "== isolated dxs=0 zero=4096"; UR_DXS=0 UR_ZERO_PAGE_BYTES=4096 python tools/pp_ablate_time.py 9,5 | grep -E "conv|gemm"
echo "== isolated dxs=0 zero=1M"; UR_DXS=0 python tools/pp_ablate_time.py 9,5 | grep -E "conv|gemm"
echo "== isolated dxs=1 zero=1M"; UR_DXS=1 python tools/pp_ablate_time.py 9,5 | grep -E "conv|gemm"
} > gpurun_out/r04/zero_dxs_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/zero_dxs_ab.txt | sed -E 's/\{"metric".*"value": ([0-9.]+),.*/  \1 steps\/s/'
