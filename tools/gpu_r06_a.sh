#!/usr/bin/env bash
# round 6 (a): counter evidence for att8::attn8_kernel (config #5's dominant kernel): the e4m3 mode of bench.py at i2v 720p and at t2v 480p
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/r06_sq_counter_names.txt
timeout 1500 bash tools/gpu_prof_r06.sh i2v720_fp8 86400 -- --model 14b-i2v --height 720 --width 1280 --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/r06_prof_i2v720_fp8.log 2>&1
tail -25 gpurun_out/r06_prof_i2v720_fp8.log
timeout 900 bash tools/gpu_prof_r06.sh t2v480_fp8 37440 -- --model 14b --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/r06_prof_t2v480_fp8.log 2>&1
tail -25 gpurun_out/r06_prof_t2v480_fp8.log
