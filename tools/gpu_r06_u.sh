#!/usr/bin/env bash
# round 6 (u): the bench lines at the round's final HEAD (no profiler)
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_bench_14b.err | tee gpurun_out/r06_final_bench_14b_driver_command_20_steps.json | cut -c1-300
timeout 600 python bench.py --model 1.3b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_final_bench_1p3b.json | cut -c1-200
timeout 600 python bench.py --gemm-dtype fp8 --attn-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_final_bench_14b_fp8_mode.json | cut -c1-200
timeout 900 python bench.py --model 14b-i2v --height 720 --width 1280 --gemm-dtype fp8 --attn-dtype fp8 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_final_bench_14b_i2v720_fp8_mode.json | cut -c1-200
