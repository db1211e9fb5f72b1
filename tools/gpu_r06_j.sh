#!/usr/bin/env bash
# round 6 (j): fuzz incl. the arrival-driven attention, the attention alone at the shard shapes, the pool-as-client record, the 50-step parity runs
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
FUZZ_R6=1 FUZZ_R5=1 FUZZ_FP8=1 timeout 400 python tools/fuzz_kernels.py 240 6 2>&1 | tail -8 | tee gpurun_out/r06_fuzz_kernels.txt
timeout 600 python tools/sp_attn_shapes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_sp_attn_shapes.txt
timeout 600 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "dead_peer" -s 2>&1 | tail -10 | tee gpurun_out/r06_dead_peer_tests.txt
ICV_TEST_SHARE_GPU=1 ICV_DIST_BACKEND=gloo ICV_WORLD=4 MODEL=small STEPS=2 timeout 900 python tools/e2e_wallclock.py 2> gpurun_out/r06_pool_e2e_rehearsal.err | tail -1 > gpurun_out/r06_pool_e2e_rehearsal.json; tail -c 700 gpurun_out/r06_pool_e2e_rehearsal.json; echo; tail -3 gpurun_out/r06_pool_e2e_rehearsal.err
timeout 900 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "worker_pool" -s 2>&1 | grep -v "MIOpen(HIP)" | tail -14 | tee gpurun_out/r06_pool_client_tests.txt
ICV_SLOW_TESTS=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "50_steps and 1.3b" 2>&1 | grep -v "MIOpen(HIP)" | tail -8 | tee gpurun_out/r06_parity_config2_50.txt
ICV_SLOW_TESTS=2 ICV_ADOPT_UNKEYED_ORACLE=1 timeout 2400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "50_steps and 14b" 2>&1 | grep -v "MIOpen(HIP)" | tail -8 | tee gpurun_out/r06_parity_config3_50.txt
