#!/usr/bin/env bash
# round 6 (c): the arrival-driven attention (csrc/attn7p.hip): kernel tests, the transports after the bounded-wait rewrite, the pool test, the compute-only projection
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attn_pieces_gpu.py -q -x -s 2>&1 | tail -30 | tee gpurun_out/r06_attn_pieces_tests.txt
timeout 1500 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "copy_engine or worker_pool_client or e4m3_on_the_wire" -s 2>&1 | grep -v "MIOpen(HIP)" | tail -30 | tee gpurun_out/r06_ipc_tests.txt
timeout 900 python -m pytest tests/test_dit_gpu.py -m gpu -q -k "ipc or copy_engine or one_rank" 2>&1 | tail -8 | tee gpurun_out/r06_dit_ipc_tests.txt
timeout 1200 python tools/sp_shard_compute_time.py 2>&1 | grep -v amdgpu.ids | tail -24 | tee gpurun_out/r06_sp_compute_only_projection.txt
