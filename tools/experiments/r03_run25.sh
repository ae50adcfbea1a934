#!/bin/bash
export AB_ATTN_SCALE0=1
echo "== A: HEAD library"; (cd gpurun_ab && AB_ATTN_ONLY=8,8,4096,4096,40 python tools/ab_attn.py 2>&1 | tail -1 | cut -c60-400)
cd uni_renderer_amd/csrc
for w in 4 3; do
  sed -i "s/__global__ void __launch_bounds__(256, [0-9]) attention32_kernel/__global__ void __launch_bounds__(256, $w) attention32_kernel/" attention.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -c attention.hip -o attention.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o norm.o attention.o misc.o backward.o attention_bwd.o tchain.o wsconv.o -o ../liburhip.so
  echo "== B: pipelined, $w waves/SIMD"
  (cd ../.. && AB_ATTN_ONLY=8,8,4096,4096,40 python tools/ab_attn.py 2>&1 | tail -1 | cut -c60-400; AB_ATTN_ONLY=1,8,16384,16384,40 python tools/ab_attn.py 2>&1 | tail -1 | cut -c60-400)
done
cd ../..
unset AB_ATTN_SCALE0
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -2
