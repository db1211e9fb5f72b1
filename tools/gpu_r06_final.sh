#!/usr/bin/env bash
# round 6, final validation at the round's HEAD: smoke, the -m gpu suite, the driver's bench command, the profiled bench, other bench lines,
# the e2e wall-clock, and a 1-GPU rehearsal of tools/first_contact_multigpu.sh (4 ranks sharing the GPU, model `small`: a code-path check)
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
STAGE=${1:-all}
if [[ $STAGE == all || $STAGE == suite ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r06_smoke.txt
  timeout 3000 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v "MIOpen(HIP)" | tail -45 | tee gpurun_out/r06_gpu_suite_summary.txt
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_bench_14b.err | tee gpurun_out/r06_bench_14b_driver_command_20_steps.json
  timeout 900 bash tools/gpu_prof_r05.sh 14b > gpurun_out/r06_prof.log 2>&1; tail -16 gpurun_out/r06_prof.log
  timeout 600 python bench.py --model 1.3b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_bench_1p3b.json
  timeout 600 python bench.py --gemm-dtype fp8 --attn-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_bench_14b_fp8_mode.json
fi
if [[ $STAGE == all || $STAGE == e2e ]]; then
  MODEL=14b STEPS=50 timeout 1500 python tools/e2e_wallclock.py 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | tee gpurun_out/r06_e2e_14b.log
  tail -1 gpurun_out/r06_e2e_14b.log > gpurun_out/r06_e2e_generate_14b.json
fi
if [[ $STAGE == all || $STAGE == rehearsal ]]; then
  ICV_BENCH_SHARE_GPU=1 MODEL=small TRACE_MODEL=small EXTRA_BENCH_ARGS="--frames 17 --height 128 --width 160" STEPS=2 SKIP_RCCL_TESTS=1 ICV_GUARD_BUDGETS="autotune=300" \
    timeout 2400 bash tools/first_contact_multigpu.sh 4 r06/first_contact_rehearsal_4ranks_sharing_one_gpu 2>&1 | tail -60 | tee gpurun_out/r06_first_contact_rehearsal.log
  mkdir -p gpurun_out/first_contact_rehearsal; cp profiles/r06/first_contact_rehearsal_4ranks_sharing_one_gpu/* gpurun_out/first_contact_rehearsal/ 2>/dev/null
fi
