#!/usr/bin/env bash
# round 6 (l): the -m gpu suite at the last HEAD + the worker-pool e2e path of the first-contact script rehearsed on one GPU (4 ranks sharing it)
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
ICV_TEST_SHARE_GPU=1 ICV_DIST_BACKEND=gloo ICV_WORLD=4 MODEL=small STEPS=2 timeout 900 python tools/e2e_wallclock.py 2> gpurun_out/r06_pool_e2e_rehearsal.err | tail -1 > gpurun_out/r06_pool_e2e_rehearsal.json; tail -c 900 gpurun_out/r06_pool_e2e_rehearsal.json; echo; grep -v "frame #" gpurun_out/r06_pool_e2e_rehearsal.err | tail -4 | cut -c1-300
timeout 3000 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v "MIOpen(HIP)" | tail -40 | tee gpurun_out/r06_gpu_suite_summary.txt
