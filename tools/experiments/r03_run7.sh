#!/bin/bash
for K in 2880 5760; do for t in 9 22 42 32; do python tools/one_gemm.py 16384 320 $K --taps 9 --tile $t --streams 2 --iters 40 2>&1 | tail -1; done; done
