#!/bin/bash
# Per-kernel totals of the captured step under rocprofv3 for two libraries (base: gpurun_ab/liburhip_base.so, tree): which kernels moved?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for tag in base tree; do
  if [ $tag = base ]; then export UR_LIB_PATH=$R/gpurun_ab/liburhip_base.so; else unset UR_LIB_PATH; fi
  rm -rf /tmp/rp_$tag
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag -o s --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-loop --no-live-traffic --no-roofline > /dev/null 2>&1)
  cp $(find /tmp/rp_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_reduce_prof_$tag.csv
done
python3 - <<PY
import csv
def load(f):
    return {r['Name']:(int(r['Calls']),float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a=load('$R/gpurun_out/r06_reduce_prof_base.csv'); b=load('$R/gpurun_out/r06_reduce_prof_tree.csv')
ta=sum(v[1] for k,v in a.items() if 'ur' in k[:12]); tb=sum(v[1] for k,v in b.items() if 'ur' in k[:12])
print('sum of our kernels: base %.2f ms  tree %.2f ms'%(ta/1e6,tb/1e6))
rows=sorted(set(a)|set(b), key=lambda n:-abs(b.get(n,(0,0))[1]-a.get(n,(0,0))[1]))
for n in rows[:14]:
    ca,da=a.get(n,(0,0)); cb,db=b.get(n,(0,0))
    print('%-90s base %5d %8.3f ms | tree %5d %8.3f ms | diff %+.3f'%(n[:90],ca,da/1e6,cb,db/1e6,(db-da)/1e6))
PY
