#!/bin/bash
# round 4, run 36: attention backward with the transposed fragments from LDS transpose reads (no q^T / k^T / dO^T copies): parity,
# then the graphed training step against a library rebuilt with -DUR_ATTN_BWD_TRN=1 (the former transposed-tile kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_backward_gpu.py -x -q -k "attention or flash" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do echo "tr reads"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200; done
cp uni_renderer_amd/liburhip.so /tmp/liburhip_new.so
(cd uni_renderer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-mfma-vgpr-form -DUR_ATTN_BWD_TRN=1 -c attention_bwd.hip -o attention_bwd.o && make 2>&1 | tail -1)
for i in 1 2; do echo "transposed tiles (UR_ATTN_BWD_TRN=1 build)"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200; done
cp /tmp/liburhip_new.so uni_renderer_amd/liburhip.so
echo "tr reads again"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200
