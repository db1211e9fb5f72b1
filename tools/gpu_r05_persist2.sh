#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/gemm_persistent_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_persist_ab_kernels_v2.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -3 | tee gpurun_out/r05_persist_tests_v2.txt
