#!/bin/bash
# wsconv: does staggering the four waves' LDS-DMA requests help?  (rebuilds wsconv.o on the box)
cd uni_renderer_amd/csrc
for sk in 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWS_SKEW=$sk -c wsconv.hip -o wsconv.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o norm.o attention.o misc.o backward.o attention_bwd.o tchain.o wsconv.o -o ../liburhip.so
  echo "== WS_SKEW=$sk"
  (cd ../.. && timeout 300 python tools/wsconv_bench.py --iters 20 --only 0,1,3,8,9,14,15 2>&1 | grep -v amdgpu.ids)
done
