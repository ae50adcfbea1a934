# L2 hit / miss counters of every kernel class over whole dual-stream steps (in situ, unlike tools/pmc_l2.sh which
# replays one problem): rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ around bench.py, reduced per kernel class with the
# split-K launches of a symbol filed separately (tools/pmc_traffic.py rules).  Output: gpurun_out/pmc_l2_step.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ --kernel-trace -d /tmp/ur_pmc_l2s -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-loop --no-live-traffic > /dev/null 2>&1)
cd $R && python - <<'PY'
import glob, json, sys
sys.path.insert(0, "tools")
import pmc_traffic as P
f = glob.glob("/tmp/ur_pmc_l2s/**/*counter_collection.csv", recursive=True)[0]
hit, miss, req = (P.per_class(f, c) for c in ("TCC_HIT", "TCC_MISS", "TCC_REQ"))
out = {}
for k in sorted(hit):
    n = max(hit[k][1], 1)
    h, m, r = hit[k][0] / n, miss[k][0] / n, req[k][0] / n
    out[k] = dict(launches=hit[k][1], tcc_hit=round(h), tcc_miss=round(m), tcc_req=round(r), hit_rate=round(h / max(h + m, 1), 4),
                  miss_bytes_128B=round(m * 128))
json.dump(out, open("gpurun_out/pmc_l2_step.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["miss_bytes_128B"])[:14]:
    print(f"{k:36s} n={v['launches']:4d} hit {v['hit_rate']:.3f} miss {v['miss_bytes_128B'] / 1e6:7.1f} MB/launch req {v['tcc_req'] * 128 / 1e6:8.1f} MB")
PY
rm -rf /tmp/ur_pmc_l2s
