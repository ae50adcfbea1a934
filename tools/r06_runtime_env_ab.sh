#!/bin/bash
# HIP runtime knobs against the captured step (graph replays): does any of them shorten the kernel boundary inside a graph?
cd ${GRAFT_REPO_ROOT:-.}
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-loop --no-live-traffic --no-roofline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run X=1)"
  echo "HIP_FORCE_DEV_KERNARG=1: $(run HIP_FORCE_DEV_KERNARG=1)"
  echo "HIP_FORCE_DEV_KERNARG=0: $(run HIP_FORCE_DEV_KERNARG=0)"
  echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1: $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1)"
  echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)"
  echo "AMD_OPT_FLUSH=0: $(run AMD_OPT_FLUSH=0)"
  echo "AMD_OPT_FLUSH=1: $(run AMD_OPT_FLUSH=1)"
  echo "DEBUG_HIP_GRAPH_BATCH_SIZE=1024: $(run DEBUG_HIP_GRAPH_BATCH_SIZE=1024)"
  echo "ROC_SYSTEM_SCOPE_SIGNAL=0: $(run ROC_SYSTEM_SCOPE_SIGNAL=0)"
  echo "GPU_MAX_HW_QUEUES=1: $(run GPU_MAX_HW_QUEUES=1)"
  echo "DEBUG_HIP_FORCE_GRAPH_QUEUES=1: $(run DEBUG_HIP_FORCE_GRAPH_QUEUES=1)"
done
