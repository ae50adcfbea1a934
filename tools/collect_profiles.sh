#!/bin/bash
# Re-create everything under profiles/ for the current build (run on the GPU box through gpurun; results land in
# gpurun_out/prof_rNN/, copy what you want judged into profiles/).  Usage: bash tools/collect_profiles.sh r01b
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
RAW=/tmp/ur_raw_$TAG   # raw traces / counter dumps are tens of MB: never under gpurun_out/ (64 MiB copy-back limit)
mkdir -p $OUT $RAW
cd $R
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-loop --no-live-traffic --shape-table $OUT/per_shape_eager_events.json > /dev/null 2>&1
export TMPDIR=/tmp
(cd $R && timeout 420 rocprofv3 --kernel-trace --stats -d $RAW/stats -o $TAG --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-loop --no-live-traffic > $OUT/bench_under_rocprofv3.json 2> $OUT/rocprof_stats.err)
(cd $R && timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pmc_fetch -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loop --no-live-traffic > /dev/null 2>&1)
(cd $R && timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pmc_write -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loop --no-live-traffic > /dev/null 2>&1)
(cd $R && timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $RAW/pmc_mfma -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loop --no-live-traffic > /dev/null 2>&1)
python tools/pmc_traffic.py $(find $RAW/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $RAW/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
python - <<PY
import csv, collections, glob, json
f = glob.glob("$RAW/pmc_mfma/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {}
for k, v in agg.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0: continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out[k] = dict(launches=n[k], mfma_util=round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4),
                  note="SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs); busy = 16 cyc per 16x16x32, 32 per 32x32x16")
json.dump(out, open("$OUT/pmc_mfma_util.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"])[:12]: print(round(v["mfma_util"], 3), v["launches"], k[:90])
PY
timeout 420 bash $R/tools/pmc_l2_step.sh > $OUT/pmc_l2_step.log 2>&1; cp $R/gpurun_out/pmc_l2_step.json $OUT/pmc_l2_step.json
cd $R
# keep only the summaries (the raw kernel traces / counter dumps are tens of MB)
cp $(find $RAW/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $RAW
ls -la $OUT
tail -2 $OUT/bench_default.json | cut -c1-400
