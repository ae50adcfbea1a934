#!/bin/bash
# round 4, run 13: where does the dx-tap-sharing kernel lose its byte advantage?  Timing-only ablation builds of igemm_dxs.hip
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
{
echo "== UR_DXS=0 (lock-step kernels)"; UR_DXS=0 python tools/pp_ablate_time.py 9 | grep -E "conv|gemm"; UR_DXS=0 $B | cut -c1-120
echo "== UR_DXS=1 product"; python tools/pp_ablate_time.py 9 | grep -E "conv|gemm"; $B | cut -c1-120
for a in 1 2 4 7; do echo "== ablation $a"; UR_LIB_PATH=$PWD/gpurun_ab/liburhip_dxsabl$a.so python tools/pp_ablate_time.py 9 | grep -E "conv|gemm"; UR_LIB_PATH=$PWD/gpurun_ab/liburhip_dxsabl$a.so $B | cut -c1-120; done
} > gpurun_out/r04/dxs_ablate.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/dxs_ablate.txt | sed -E 's/\{"metric".*"value": ([0-9.]+),.*/  \1 steps\/s/'
