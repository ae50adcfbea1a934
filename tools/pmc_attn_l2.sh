#!/bin/bash
# L2 / fabric behaviour of the d = 40 self-attention at cfg 5's level-0 shape (T = 16384; 2 x 8 heads = both streams of one sample):
# hit rate, request sizes on the memory side, FETCH_SIZE / WRITE_SIZE -- is the "2.46x algorithmic" traffic of the bench line re-streamed
# K / V^T, or 80-byte head slices of 640-byte token rows (sector over-fetch), or the x2 correction of FETCH_SIZE applied to requests that
# are NOT 128-byte reads (MI355X_MICROARCH.md: the x2 is calibrated for wide coalesced streams only)?  Separate passes, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_ATTN_ONLY=${AB_ATTN_ONLY:-2,8,16384,16384,40}
i=0
for set in "TCC_HIT TCC_MISS TCC_REQ TCC_READ" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_WRITE"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_attn_l2/s$i -o p --output-format csv -- python tools/ab_attn.py > /tmp/pmc_attn_l2_s$i.log 2>&1; tail -2 /tmp/pmc_attn_l2_s$i.log >&2)
done
cd $R && python - <<'PY'
import csv, collections, glob, json, os
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_attn_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention" in r["Kernel_Name"] and "ur" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k,v in agg.items()}
B,H,T,Tk,d=(int(v) for v in os.environ["AB_ATTN_ONLY"].split(","))
alg=4*B*H*T*d*2   # q, k, v, o once, dense
out=dict(problem=dict(B=B,H=H,T=T,Tk=Tk,d=d), counters_per_launch={k: round(v) for k,v in sorted(m.items())}, algorithmic_bytes=alg)
if "TCC_HIT" in m and "TCC_MISS" in m: out["l2_hit_rate"]=round(m["TCC_HIT"]/(m["TCC_HIT"]+m["TCC_MISS"]),4)
if "TCC_EA0_RDREQ" in m:
    r32=m.get("TCC_EA0_RDREQ_32B",0.0)
    out["memory_side_read_bytes_if_all_requests_were_64B"]=round(m["TCC_EA0_RDREQ"]*64)
    out["memory_side_read_bytes_32B_requests_counted_as_32B"]=round((m["TCC_EA0_RDREQ"]-r32)*64+r32*32)
    out["share_of_32B_read_requests"]=round(r32/max(m["TCC_EA0_RDREQ"],1),4)
if "FETCH_SIZE" in m: out["FETCH_SIZE_KB"]=round(m["FETCH_SIZE"]); out["bench_traffic_formula_bytes (FETCH x 2 x 1024 + WRITE x 1024)"]=round(m["FETCH_SIZE"]*2048+m.get("WRITE_SIZE",0)*1024)
print(json.dumps(out, indent=1))
PY
rm -rf /tmp/pmc_attn_l2
