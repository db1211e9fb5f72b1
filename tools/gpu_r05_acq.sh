#!/bin/bash
# validation of the single-launch icv_ipc_acquire (wait_done_kernel) on the GPU
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dit_gpu.py -q -m gpu -k "copy_engine or misuse" 2>&1 | tail -5 > gpurun_out/r05_acq_dit.txt
timeout 1500 python -m pytest tests/test_multigpu_rccl.py -q -m gpu -s -k "ipc or copy_engine or fp8 or 14b_layer" 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r05_acq_mg.txt
cat gpurun_out/r05_acq_dit.txt gpurun_out/r05_acq_mg.txt
