#!/usr/bin/env bash
# round 6 (f): transports after the bounded-wait rewrite, arrival-driven attention in the DiT (one rank, 2-4 processes sharing the GPU), worker pool as client, >4 GiB conv
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dit_gpu.py -m gpu -q -k "ipc or copy_engine or one_rank or arrival" 2>&1 | tail -8 | tee gpurun_out/r06_dit_ipc_tests.txt
timeout 2400 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "copy_engine or worker_pool or e4m3_on_the_wire or arrival" -s 2>&1 | grep -v "MIOpen(HIP)" | tail -40 | tee gpurun_out/r06_ipc_tests.txt
timeout 900 python -m pytest tests/test_vae_hip_gpu.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r06_vae_hip_tests.txt
