R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { UR_EXPERIMENT=$1 timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 200 --direction render --batch 2 --latent 32 --dtype bf16 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for rep in 1 2 3; do for e in "" "splitk_gn"; do echo "cfg2 [$e] $(one "$e")"; done; done
