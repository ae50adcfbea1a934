#!/bin/bash
# SQ-counter breakdown of ONE igemm problem replayed alone ("find the conv's real limiter with
# counters").  Default: the level-0 conv of the headline step (M = 2 x 16384, N = 320, K = 2880, tile 128x320).
# Three separate passes (8 SQ slots each; counters never combined with tracing beyond --kernel-trace).
#   bash tools/pmc_conv_sq.sh [out.json]       ARGS="16384 320 2880 --taps 9 --tile 9 --streams 2 --iters 20"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc_conv_sq.json}
ARGS=${ARGS:-"16384 320 2880 --taps 9 --tile 9 --streams 2 --iters 20"}
D=$R/gpurun_out/pmc_conv_sq
rm -rf $D
mkdir -p $R/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $set --kernel-trace -d $D/s$i -o p --output-format csv -- python tools/one_gemm.py $ARGS > $D.log$i 2>&1)
done
cd $R && python - "$OUT" "$ARGS" <<'PY'
import csv, collections, glob, json, sys
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_conv_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
out = dict(problem=sys.argv[2], per_launch=m, n_launches={k: len(v) for k, v in agg.items()})
w = m.get("SQ_WAVE_CYCLES", 0.0)
if w:
    out["fractions_of_SQ_WAVE_CYCLES"] = {k: round(m[k] / w, 4) for k in
        ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
         "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA") if k in m}
if m.get("GRBM_GUI_ACTIVE"):
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    out["kernel_cycles"] = cyc
    out["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024.0), 4)
    out["lds_idx_active_frac_per_cu"] = round(m.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256.0), 4)
    out["lds_bank_conflict_frac_of_lds_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
rm -rf $D
