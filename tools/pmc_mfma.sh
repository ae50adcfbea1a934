cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_mfma_ub -o p --output-format csv -- $R/tools/ubench/mfma_rate > /dev/null 2>&1
cd $R && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_mfma_attn -o p --output-format csv -- python tools/ab_attn.py > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("gpurun_out/pmc_mfma_ub","gpurun_out/pmc_mfma_attn"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            print(d.split('/')[-1], k, {c:(round(sum(x)/len(x)), len(x)) for c,x in v.items()})
PY
