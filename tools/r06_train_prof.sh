cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 500 python tools/train_bench.py --graph --steps 3 > gpurun_out/r06_train_graph_base.json 2>/dev/null; cat gpurun_out/r06_train_graph_base.json | head -c 600; echo
timeout 500 python tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype bf16 --algorithm rs_ag 2>/dev/null | head -1 > gpurun_out/r06_train_graph_rccl_base.json; cat gpurun_out/r06_train_graph_rccl_base.json | head -c 900; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ur_tr -o tr --output-format csv -- python $GRAFT_REPO_ROOT/tools/train_bench.py --graph --steps 3 --force-collectives --comm-dtype bf16 --algorithm rs_ag > /dev/null 2>&1)
cp $(find /tmp/ur_tr -name "*kernel_stats.csv" | head -1) gpurun_out/r06_train_rccl_kernel_stats_base.csv
head -30 gpurun_out/r06_train_rccl_kernel_stats_base.csv | cut -c1-150
