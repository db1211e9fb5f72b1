#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_attn_pieces_gpu.py tests/test_dit_gpu.py -m gpu -q -x 2>&1 | grep -v "RCCL\|ROCm version\|Hostname\|Librccl\|amdgpu.ids\|HIP version" | tail -6
FUZZ_R6=1 timeout 200 python tools/fuzz_kernels.py 90 11 2>&1 | tail -3
bash tools/gpu_r06_final.sh bench
