#!/bin/bash
# round 4, run 40: pack / unpack of the conv weights through LDS (coalesced on both sides): parity, then the graphed step against
# the previous commit's kernels rebuilt on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "pack_unpack" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -k "graph or oracle_autograd or sd_size" 2>&1 | tail -2
for i in 1 2; do echo "LDS-staged pack / unpack"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160; done
cp uni_renderer_amd/liburhip.so /tmp/liburhip_new.so
cp uni_renderer_amd/csrc/backward.hip /tmp/backward_new.hip
cp tools/experiments/backward_prev.hip uni_renderer_amd/csrc/backward.hip
(cd uni_renderer_amd/csrc && make 2>&1 | tail -1)
for i in 1 2; do echo "previous pack / unpack kernels"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160; done
cp /tmp/backward_new.hip uni_renderer_amd/csrc/backward.hip; cp /tmp/liburhip_new.so uni_renderer_amd/liburhip.so
echo "LDS-staged again"; python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160
