#!/bin/bash
# does a 3-deep LDS ring help a big conv tile?  (128x256, N = 256 problem of the level-0 shape)
for t in 10 41; do python tools/one_gemm.py 16384 256 2880 --taps 9 --tile $t --streams 2 --iters 40 2>&1 | tail -1; done
for t in 10 41; do python tools/one_gemm.py 16384 256 5760 --taps 9 --tile $t --streams 2 --iters 40 2>&1 | tail -1; done
for t in 10 41 9; do python tools/one_gemm.py 16384 256 1280 --tile $t --iters 40 2>&1 | tail -1; done
