#!/bin/bash
# Round 6: memory round trips of the chain kernels (csrc/tchain.hip): 10 loads in flight per lane instead of 5 (variant a), and no
# store drain before the ring refills (variant b = the tree).  Libraries gpurun_ab/liburhip_{prev,a_inflight10,b_nodrain}.so,
# alternating on one box: the three chains in isolation (tools/tchain_bench.py) and the headline step / cfg 5.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_tchain_gpu.py -x -q > $O/r06_tchain_io_tests.log 2>&1; tail -1 $O/r06_tchain_io_tests.log
one() { timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 $@ 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
{
for lib in prev a_inflight10 b_nodrain; do
  echo "$lib: $(UR_LIB_PATH=$R/gpurun_ab/liburhip_$lib.so timeout 300 python tools/tchain_bench.py 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("fused_pre_us","fused_q_us","fused_ff_us","profile_q","profile_ff")})')"
done
for rep in 1 2 3; do
  for lib in prev a_inflight10 b_nodrain; do echo "cfg3 [$lib] $(UR_LIB_PATH=$R/gpurun_ab/liburhip_$lib.so one)"; done
done
for rep in 1 2; do
  for lib in prev b_nodrain; do echo "cfg5 [$lib] $(UR_LIB_PATH=$R/gpurun_ab/liburhip_$lib.so one --batch 1 --latent 128)"; done
done
} > $O/r06_tchain_io_ab.txt 2>&1
cat $O/r06_tchain_io_ab.txt
timeout 600 python tools/tchain_determinism.py > $O/r06_tchain_determinism.txt 2>&1; tail -3 $O/r06_tchain_determinism.txt
