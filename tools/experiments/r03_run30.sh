#!/bin/bash
timeout 900 python -m pytest tests/test_wsconv_gpu.py -x -q 2>&1 | tail -4
for wv in 4 8; do echo "== UR_WSCONV_WAVES=$wv"; UR_WSCONV_WAVES=$wv timeout 600 python tools/wsconv_bench.py --iters 20 --only 0,1,3,8,9,14,15 2>&1 | grep -v amdgpu.ids; done
