#!/bin/bash
# which of the two debug knobs breaks the debug build at act = 0?
cd uni_renderer_amd/csrc
for dbg in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWS_DEBUG=$dbg -c wsconv.hip -o wsconv.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o norm.o attention.o misc.o backward.o attention_bwd.o tchain.o wsconv.o -o ../liburhip.so
  echo "== WS_DEBUG=$dbg"
  (cd ../.. && timeout 300 python -m pytest tests/test_wsconv_gpu.py -q 2>&1 | tail -3)
done
