#!/bin/bash
# Round 6: head-major q / k images for the d = 40 attention (fused.HEAD_MAJOR_QK).  Tests, the L2 counters of the cfg-5 attention
# with both layouts, and the step with / without the images alternating on one box (headline cfg 3 and cfg 5).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_tchain_gpu.py tests/test_golden_sd_gpu.py tests/test_parity_sweep_gpu.py -x -q > $O/r06_head_major_tests.log 2>&1; tail -2 $O/r06_head_major_tests.log
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k attention >> $O/r06_head_major_tests.log 2>&1; tail -1 $O/r06_head_major_tests.log
AB_ATTN_HEAD_MAJOR=1 bash tools/pmc_attn_l2.sh > $O/r06_pmc_attn_l2_cfg5_head_major.json 2> $O/r06_pmc_attn_l2_hm.err
bash tools/pmc_attn_l2.sh > $O/r06_pmc_attn_l2_cfg5.json 2>> $O/r06_pmc_attn_l2_hm.err
{
for rep in 1 2 3; do
  for e in "" "no_head_major_qk"; do
    echo "cfg3 UR_EXPERIMENT=$e: $(UR_EXPERIMENT=$e timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  done
done
for rep in 1 2; do
  for e in "" "no_head_major_qk"; do
    echo "cfg5 UR_EXPERIMENT=$e: $(UR_EXPERIMENT=$e timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --batch 1 --latent 128 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"])')"
  done
done
AB_ATTN_ONLY=2,8,16384,16384,40 AB_ATTN_SCALE0=1 python tools/ab_attn.py 2>/dev/null | tail -1
AB_ATTN_ONLY=2,8,16384,16384,40 AB_ATTN_SCALE0=1 AB_ATTN_HEAD_MAJOR=1 python tools/ab_attn.py 2>/dev/null | tail -1
AB_ATTN_ONLY=8,8,4096,4096,40 AB_ATTN_SCALE0=1 python tools/ab_attn.py 2>/dev/null | tail -1
AB_ATTN_ONLY=8,8,4096,4096,40 AB_ATTN_SCALE0=1 AB_ATTN_HEAD_MAJOR=1 python tools/ab_attn.py 2>/dev/null | tail -1
} > $O/r06_head_major_ab.txt 2>&1
cat $O/r06_head_major_ab.txt
