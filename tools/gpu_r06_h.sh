#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -q -k "arrival or native_forward" 2>&1 | grep -v "RCCL\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8 | tee gpurun_out/r06_dit_arrival_tests.txt
for layout in sp auto; do
  ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo ICV_GUARD_BUDGETS="autotune=300" timeout 1500 python bench.py --gpus 4 --parallelism $layout --model small --frames 17 --height 128 --width 160 --steps 3 --warmup 1 --no-cpu-baseline \
    2> gpurun_out/r06_bench_4ranks_${layout}_shared.err > gpurun_out/r06_bench_selflaunch_4ranks_${layout}_shared_gpu_gloo.json
  grep "\[bench\] autotune\|FAILED" gpurun_out/r06_bench_4ranks_${layout}_shared.err | cut -c1-400
done
timeout 2400 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "bench" 2>&1 | grep -v "MIOpen(HIP)" | tail -12 | tee gpurun_out/r06_bench_line_tests.txt
