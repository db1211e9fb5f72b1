#!/usr/bin/env bash
# round 5, lease A: the copy-engine K|V transport's GPU tests + the contention table
set -uo pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_dit_gpu.py -k "copy_engine" -x -q -s 2>&1 | tail -30 | tee gpurun_out/r05a_ipc_one_rank.txt
timeout 900 python -m pytest tests/test_multigpu_rccl.py -k "copy_engine" -x -q -s 2>&1 | tail -40 | tee gpurun_out/r05a_ipc_shared_gpu.txt
timeout 900 python tools/kv_contention.py 2>&1 | tee gpurun_out/kv_contention.txt
NCCL_MAX_NCHANNELS=2 WHAT=rccl timeout 300 python tools/kv_contention.py 2>&1 | tee gpurun_out/kv_contention_rccl2.txt
