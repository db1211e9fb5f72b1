#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "copy_path_probe" 2>&1 | tail -6 | tee gpurun_out/r06_copy_probe_controls.txt
bash tools/gpu_r06_final.sh bench
