#!/bin/bash
# round 3, GPU call 1: MFMA issue-rate ubench (fixed), SQ counters of the level-0 conv, baseline bench line
mkdir -p gpurun_out
./tools/ubench/mfma_rate > gpurun_out/r03_mfma_rate.txt 2>&1; cat gpurun_out/r03_mfma_rate.txt
bash tools/pmc_conv_sq.sh gpurun_out/r03_pmc_conv_sq.json > gpurun_out/r03_pmc_conv_sq.log 2>&1; tail -60 gpurun_out/r03_pmc_conv_sq.log
ARGS="16384 320 320 --tile 9 --streams 1 --iters 50 --res" bash tools/pmc_conv_sq.sh gpurun_out/r03_pmc_gemm320_sq.json > gpurun_out/r03_pmc_gemm320_sq.log 2>&1; tail -40 gpurun_out/r03_pmc_gemm320_sq.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err; cut -c1-600 gpurun_out/r03_bench0.json
