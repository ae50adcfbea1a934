#!/bin/bash
# round 4, run 39: packed conv weight gradient -> fp32 [Co, Ci, 3, 3] with coalesced stores through LDS: parity, kernel time, step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "pack_unpack" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -k "graph or oracle_autograd or adamw or sd_size" 2>&1 | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from uni_renderer_amd import _lib
from uni_renderer_amd.ops import DT
lib = _lib.load()
for co, ci in ((1280, 2560), (640, 640), (320, 320), (1280, 1280)):
    dwp = torch.randn(co, 9 * ci, device='cuda').to(torch.bfloat16)
    g = torch.empty(co, ci, 3, 3, device='cuda')
    part = torch.empty(int(lib.ur_unpack_conv_weight_grad_blocks(co, ci)), device='cuda')
    f = lambda: lib.ur_unpack_conv_weight_grad_sumsq(dwp.data_ptr(), dwp.stride(0), g.data_ptr(), co, ci, ci, part.data_ptr(), DT[dwp.dtype], torch.cuda.current_stream().cuda_stream)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"unpack {co}x{ci}: {us:.1f} us, {co * ci * 9 * 6 / us / 1e6:.2f} TB/s")
PY
for i in 1 2; do python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-160; done
