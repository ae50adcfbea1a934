#!/bin/bash
# round 4, run 11: (a) timing-only UPPER BOUND of "the three dx taps share one staged pixel block": a build of igemm.hip that
# copies the pixel tile for one tap in three (-DUR_ABLATE=3, results are garbage) -- isolated and in the step;
# (b) one-launch GroupNorm up to 1024 rows (the 32x32 level) now that its groups sit on one XCD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
{
for i in 1 2; do
echo "product build";                      $B | cut -c1-120
echo "dx-share upper bound build";         UR_LIB_PATH=$PWD/gpurun_ab/liburhip_dxshare_bound.so $B | cut -c1-120
echo "gn fused up to 1024 rows";           UR_GN_FUSED_MAX_ROWS=1024 $B | cut -c1-120
done
echo "== isolated, product build"; python tools/pp_ablate_time.py 9,32
echo "== isolated, dx-share upper bound build"; UR_LIB_PATH=$PWD/gpurun_ab/liburhip_dxshare_bound.so python tools/pp_ablate_time.py 9,32
} > gpurun_out/r04/dxshare_bound.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/dxshare_bound.txt | sed -E 's/\{"metric".*"value": ([0-9.]+),.*/  \1 steps\/s/'
