#!/bin/bash
# Round 6: output-tile order inside an XCD's run (ur_igemm_desc.raster).  Test, then the headline step / cfg 2 / cfg 5 / the
# hoisted loop with the library's per-launch choice (default) against n fastest everywhere (igemm_raster=1, rounds 1-5),
# alternating on one box; then the library against the previous library (tools/lib_ab.sh protocol: UR_LIB_PATH) when given.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "tile_order or linear_bias_res or geglu" > $O/r06_raster_tests.log 2>&1; tail -1 $O/r06_raster_tests.log
one() { UR_EXPERIMENT=$1 timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic ${@:2} 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
{
for rep in 1 2 3; do
  for e in "" "igemm_raster=1"; do echo "cfg3 [$e] $(one "$e")"; done
done
for rep in 1 2; do
  for e in "" "igemm_raster=1"; do echo "cfg2 [$e] $(one "$e" --direction render --batch 2 --latent 32 --dtype bf16)"; done
  for e in "" "igemm_raster=1"; do echo "cfg5 [$e] $(one "$e" --batch 1 --latent 128)"; done
  for e in "" "igemm_raster=1"; do echo "b8 [$e] $(one "$e" --batch 8)"; done
done
if [ -n "$PREV_LIB" ]; then
  for rep in 1 2 3; do
    echo "cfg3 [new lib] $(one "")"
    echo "cfg3 [prev lib] $(UR_LIB_PATH=$PREV_LIB one "igemm_raster=1")"
  done
fi
} > $O/r06_raster_ab.txt 2>&1
cat $O/r06_raster_ab.txt
