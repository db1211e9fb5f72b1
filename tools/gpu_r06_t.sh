#!/usr/bin/env bash
# round 6 (t): the arrival gate of the e4m3 chunk launches: kernel test, one-rank rehearsal, real processes sharing the GPU, the e4m3 regression tests
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -m gpu -q -x -k "fp8 or e4m3" 2>&1 | grep -v "MIOpen(HIP)" | tail -15 | tee gpurun_out/r06_e4m3_gate_tests.txt
timeout 1200 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "fp8 or e4m3" -s 2>&1 | grep -v "MIOpen(HIP)\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 | tee -a gpurun_out/r06_e4m3_gate_tests.txt
