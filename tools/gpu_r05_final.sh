#!/usr/bin/env bash
# round 5, final validation at the round's HEAD: the -m gpu suite, the driver's bench command, the profiled bench, the e2e wall-clock
# (first call of a fresh process AND steady state, 50 steps each), other bench lines, the compute-only projection.
set -uo pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
STAGE=${1:-all}
if [[ $STAGE == all || $STAGE == suite ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r05_smoke.txt
  timeout 2400 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v "MIOpen(HIP)" | tail -40 | tee gpurun_out/r05_gpu_suite_summary.txt
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_bench_14b.err | tee gpurun_out/r05_bench_14b_driver_command_20_steps.json
  timeout 900 bash tools/gpu_prof_r05.sh 14b > gpurun_out/r05_prof.log 2>&1
  timeout 600 python bench.py --model 1.3b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_bench_1p3b.json
  timeout 600 python bench.py --gemm-dtype fp8 --attn-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_bench_14b_fp8_mode.json
fi
if [[ $STAGE == all || $STAGE == e2e ]]; then
  MODEL=14b STEPS=50 timeout 1500 python tools/e2e_wallclock.py 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | tee gpurun_out/r05_e2e_14b.log
  tail -1 gpurun_out/r05_e2e_14b.log > gpurun_out/r05_e2e_generate_14b.json
fi
if [[ $STAGE == all || $STAGE == proj ]]; then
  timeout 900 python tools/sp_shard_compute_time.py 2>&1 | tail -16 | tee gpurun_out/r05_sp_compute_only_projection.txt
fi
