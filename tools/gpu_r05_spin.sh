#!/bin/bash
# what the copy-engine transport's pending waits (spin waves) cost the attention they sit beside
export TMPDIR=/tmp; mkdir -p gpurun_out
for q in 16 4; do
  echo "=== GPU_MAX_HW_QUEUES=$q" | tee -a gpurun_out/r05_kv_contention_spin.txt
  GPU_MAX_HW_QUEUES=$q WHAT=spin ITERS=5 timeout 240 python -u tools/kv_contention.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_kv_contention_spin.txt
done
