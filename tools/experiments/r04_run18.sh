#!/bin/bash
# round 4, run 18: ur_wgrad ablations (library built with make WGRAD_ABL=1) + LDS conflict counters of the full kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{ for t in 1 2 3; do echo "== tile $t"; python tools/wgrad_ablate.py --tile $t 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r04/wgrad_ablate.txt
cat gpurun_out/r04/wgrad_ablate.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
UR_WGRAD_ABLATE=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/r04/pmc_wgrad -o w --output-format csv -- python tools/wgrad_ablate.py --one > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r04/pmc_wgrad/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'wgrad_kernel' in r['Kernel_Name']:
            acc[(r['Dispatch_Id'], r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']] += float(r['Counter_Value'])
with open('gpurun_out/r04/wgrad_pmc.txt', 'w') as o:
    for k, v in acc.items():
        o.write(f"{k} {dict(v)}\n")
print(open('gpurun_out/r04/wgrad_pmc.txt').read()[:3000])
PY
rm -rf gpurun_out/r04/pmc_wgrad
