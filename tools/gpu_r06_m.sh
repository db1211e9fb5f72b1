#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attn_pieces_gpu.py tests/test_dit_gpu.py -m gpu -q -k "pieces or rows_that_land or never_arrives or ipc or arrival or copy_engine" 2>&1 | grep -v "RCCL\|ROCm version\|Hostname\|Librccl\|amdgpu.ids\|HIP version" | tail -6
timeout 1800 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "copy_engine or dead_peer or publishes_late or arrival or worker_pool_client or e4m3_on_the_wire" 2>&1 | grep -v "MIOpen(HIP)" | tail -8 | tee gpurun_out/r06_transport_tests_last_head.txt
