#!/bin/bash
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('default table', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
UR_IGEMM_TUNING=tools/data/igemm_tuning_l4conv.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('L4 level-0 convs', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
