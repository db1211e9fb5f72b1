#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for w in 0 1; do
echo "=== WARM=$w" | tee -a gpurun_out/r05_kv_contention_control.txt
WARM=$w GPU_MAX_HW_QUEUES=16 WHAT=control,spin,hostfunc,occupy ITERS=5 timeout 400 python -u tools/kv_contention.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_kv_contention_control.txt
done
