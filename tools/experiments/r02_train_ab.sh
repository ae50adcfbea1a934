#!/bin/bash
# same-box alternation: this round's training-path switches off / on (the fold-kernel and host-side changes have no switch)
mkdir -p gpurun_out; : > gpurun_out/train_ab.txt
for i in 1 2; do
  ms=$(python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $i $ms" | tee -a gpurun_out/train_ab.txt
  ms=$(UR_FLASH_BACKWARD=0 UR_BATCH_CASTS=0 UR_MULTI_TRANSPOSE=0 UR_FORWARD_LSE=0 python tools/train_bench.py --steps 5 --graph --torch-adamw 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "materialised_P+torch_adamw+per_tensor_casts+single_transposes $i $ms" | tee -a gpurun_out/train_ab.txt
  ms=$(UR_FLASH_BACKWARD=0 python tools/train_bench.py --steps 5 --graph 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "only_flash_backward_off $i $ms" | tee -a gpurun_out/train_ab.txt
  ms=$(python tools/train_bench.py --steps 5 --graph --torch-adamw 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "only_torch_adamw $i $ms" | tee -a gpurun_out/train_ab.txt
done
