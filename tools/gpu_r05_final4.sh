#!/usr/bin/env bash
# round 5, the very last HEAD: -m gpu suite
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "MIOpen(HIP)" | tail -14 | tee gpurun_out/r05_gpu_suite_summary_final4.txt
