#!/usr/bin/env bash
# round 5: the -m gpu suite after the oracle's explicit attention lost one pass over the scores and icv_ipc_abort landed
set -uo pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -s -k "failing_is_an_error or copy_engine" 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r05_ipc_abort_tests.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v "MIOpen(HIP)" | tail -32 | tee gpurun_out/r05_gpu_suite_summary_head3.txt
