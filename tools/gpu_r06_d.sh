#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attn_pieces_gpu.py -q -s 2>&1 | tail -40 | tee gpurun_out/r06_attn_pieces_tests.txt
timeout 900 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -x -k "copy_engine_transport_is_bit_identical" -s 2>&1 | grep -v "MIOpen(HIP)" | tail -150 > gpurun_out/r06_ipc_fail.txt; tail -5 gpurun_out/r06_ipc_fail.txt
timeout 1200 python tools/sp_shard_compute_time.py 2>&1 | grep -v amdgpu.ids | tail -24 | tee gpurun_out/r06_sp_compute_only_projection.txt
