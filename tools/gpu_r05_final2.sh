#!/usr/bin/env bash
# round 5, last lease: the -m gpu suite at the very last HEAD; the supervised 4-rank bench on the shared GPU (code-path record with the
# copy-engine transport in the autotune table)
set -uo pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v "MIOpen(HIP)" | tail -30 | tee gpurun_out/r05_gpu_suite_summary_last_head.txt
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --model small --frames 17 --height 128 --width 160 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r05_bench_4ranks_shared.err | tee gpurun_out/r05_bench_selflaunch_4ranks_shared_gpu_gloo.json
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --parallelism sp --model small --frames 17 --height 128 --width 160 --steps 3 --warmup 1 --no-cpu-baseline 2>> gpurun_out/r05_bench_4ranks_shared.err | tee gpurun_out/r05_bench_selflaunch_4ranks_sp_shared_gpu_gloo.json
tail -30 gpurun_out/r05_bench_4ranks_shared.err
