# PMC breakdown of the d = 40 self-attention kernel (one problem, separate passes of <= 8 SQ counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_ATTN_ONLY=${AB_ATTN_ONLY:-8,8,4096,4096,40}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_attn/s$i -o p --output-format csv -- python tools/ab_attn.py > /dev/null 2>&1)
done
cd $R && python - <<'PY'
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention" in r["Kernel_Name"] and "ur" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(f"{k:32s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
