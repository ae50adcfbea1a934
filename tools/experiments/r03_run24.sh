#!/bin/bash
export AB_ATTN_SCALE0=1
echo "== A: HEAD library"; (cd gpurun_ab && AB_ATTN_ONLY=8,8,4096,4096,40 python tools/ab_attn.py 2>&1 | tail -1)
echo "== B: tree"; AB_ATTN_ONLY=8,8,4096,4096,40 python tools/ab_attn.py 2>&1 | tail -1
unset AB_ATTN_SCALE0
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -3
