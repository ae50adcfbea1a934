#!/bin/bash
mkdir -p gpurun_out
{
echo "== step determinism: latent 128, batch 1 (cfg 5 shape)"; timeout 600 python tools/determinism_check.py --latent 128 --batch 1 --n 100 2>&1 | tail -2
echo "== step determinism: latent 32, batch 2"; timeout 600 python tools/determinism_check.py --latent 32 --batch 2 --n 200 2>&1 | tail -2
echo "== step determinism: latent 64, batch 4"; timeout 600 python tools/determinism_check.py --latent 64 --batch 4 --n 300 2>&1 | tail -2
echo "== chains, M = 16384"; timeout 600 python tools/tchain_determinism.py --iters 2000 2>&1 | tail -3
echo "== chains, M = 4096 (ragged workgroup count)"; timeout 600 python tools/tchain_determinism.py --iters 2000 --M 4096 2>&1 | tail -3
echo "== chains, bf16"; timeout 600 python tools/tchain_determinism.py --iters 1000 --dtype bf16 2>&1 | tail -3
} | grep -v amdgpu.ids | tee gpurun_out/r03_determinism.txt
