# L2 behaviour of ONE igemm problem (default: the level-0 conv of the headline step, z = 2): hit / miss / EA reads.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${ARGS:-"16384 320 2880 --taps 9 --tile 9 --streams 2 --iters 20"}
i=0
for set in "TCC_HIT TCC_MISS TCC_REQ TCC_READ" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM" "TCC_STREAMING_REQ TCC_NC_READ_REQ TCC_UC_READ_REQ TCC_RW_READ_REQ"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_l2/s$i -o p --output-format csv -- python tools/one_gemm.py $ARGS > /dev/null 2>&1)
done
cd $R && python - <<'PY'
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
rm -rf $R/gpurun_out/pmc_l2
