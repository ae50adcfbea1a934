#!/bin/bash
# round 4, run 21: L2 hit / miss counters of ur_wgrad (tiles 1 and 4, three problems) and kernel durations from the trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 1 4; do
UR_WGRAD_ABLATE=0 rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ --kernel-trace -d gpurun_out/r04/pmc_wg$t -o w --output-format csv -- python tools/wgrad_ablate.py --one --tile $t > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
with open('gpurun_out/r04/wgrad_l2.txt', 'w') as o:
    for t in (1, 4):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for fn in glob.glob(f'gpurun_out/r04/pmc_wg{t}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(fn)):
                if 'wgrad_kernel' in r['Kernel_Name'] and 'ELi0EEEv' in r['Kernel_Name']:
                    acc[(r['Dispatch_Id'], r['Grid_Size'])][r['Counter_Name']] += float(r['Counter_Value'])
        dur = {}
        for fn in glob.glob(f'gpurun_out/r04/pmc_wg{t}/**/*kernel_trace.csv', recursive=True):
            for r in csv.DictReader(open(fn)):
                dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        for k, v in acc.items():
            o.write(f"tile {t} dispatch {k} dur_us {dur.get(k[0])} {dict(v)} hit_rate {v['TCC_HIT'] / max(1, v['TCC_HIT'] + v['TCC_MISS']):.3f} miss_MB {v['TCC_MISS'] * 128 / 1e6:.1f}\n")
print(open('gpurun_out/r04/wgrad_l2.txt').read())
PY
rm -rf gpurun_out/r04/pmc_wg1 gpurun_out/r04/pmc_wg4
