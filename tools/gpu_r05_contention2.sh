#!/bin/bash
# kv_contention with the baseline measured through the same harness as the co-runner rows (the first version booked the harness's idle gap on the co-runner)
export TMPDIR=/tmp; mkdir -p gpurun_out
GPU_MAX_HW_QUEUES=16 WHAT=occupy,spin,hostfunc,blit,rccl,gemm ITERS=5 timeout 600 python -u tools/kv_contention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_kv_contention_v2.txt
GPU_MAX_HW_QUEUES=16 WHAT=occupy,spin,hostfunc ITERS=5 timeout 600 python -u tools/kv_contention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_kv_contention_v2_repeat.txt
