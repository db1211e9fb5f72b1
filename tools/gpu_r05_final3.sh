#!/usr/bin/env bash
# round 5, final HEAD (persistent GEMM default, icv_ipc_abort): -m gpu suite, rocprofv3 summary of bench.py, the driver's bench command
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "MIOpen(HIP)" | tail -16 | tee gpurun_out/r05_gpu_suite_summary_final.txt
bash tools/gpu_prof_r05.sh 14b > gpurun_out/r05_prof_final.log 2>&1; tail -20 gpurun_out/r05_prof_final.log
cp gpurun_out/prof_r05_14b/stats/*/*kernel_stats.csv gpurun_out/r05_kernel_stats_final.csv 2>/dev/null || find gpurun_out/prof_r05_14b/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_kernel_stats_final.csv \;
rm -rf gpurun_out/prof_r05_14b
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_bench_final.err | tee gpurun_out/r05_bench_final.json
