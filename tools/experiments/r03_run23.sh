#!/bin/bash
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_attn
export AB_ATTN_SCALE0=1
AB_ATTN_ONLY=8,8,4096,4096,40 python tools/ab_attn.py 2>&1 | tail -2
bash tools/pmc_attn.sh 2>&1 | tee gpurun_out/r03_pmc_attn_slot.txt
rm -rf gpurun_out/pmc_attn
