#!/bin/bash
# wsconv (4 waves): placement of the weight copies inside the stage stream -- every MFMA from the start (step 1), every
# second (2), every third (3 = the first version)
cd uni_renderer_amd/csrc
for st in 1 2 3; do
  UR_GEN_CONV_WSTEP=$st python ../../tools/gen_tchain_asm.py > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c wsconv.hip -o wsconv.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o norm.o attention.o misc.o backward.o attention_bwd.o tchain.o wsconv.o -o ../liburhip.so
  echo "== weight-copy step $st"
  (cd ../.. && UR_WSCONV_WAVES=4 timeout 300 python tools/wsconv_bench.py --iters 20 --only 0,1,3,9,15 2>&1 | grep -v amdgpu.ids)
done
