#!/bin/bash
# round 4, run 33: full GPU suite on the tree with ur_wgrad / deferred gradients / the kernel body refactor; step and training timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/full_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/full_pytest.log
grep -E "passed|failed|rc=" gpurun_out/r04/full_pytest.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200; done
for i in 1 2; do python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | cut -c1-200; done
