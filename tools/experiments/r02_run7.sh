#!/bin/bash
# clocks / power while the step graph replays back to back (is the step power limited?)
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30 > gpurun_out/r7_smi_idle.txt
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r7_bench_long.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" ; echo ---; sleep 2; done > gpurun_out/r7_smi_busy.txt
wait $BP
cat gpurun_out/r7_smi_idle.txt | head -20; echo ======; head -30 gpurun_out/r7_smi_busy.txt
python -c "
import json
print(json.loads(open('gpurun_out/r7_bench_long.json').read().strip().splitlines()[-1])['ms_per_step'])"
