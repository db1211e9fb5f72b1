#!/usr/bin/env bash
# round 6 (g): arrival tests after the expectation fix; the 4-rank shared-GPU bench records with the new autotune columns; a 1-GPU rehearsal of
# tools/first_contact_multigpu.sh; the parity tests with the Wan-VAE-architecture frame PSNR (configs #2 / #3)
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -q -k "arrival or native_forward" 2>&1 | tail -5 | tee gpurun_out/r06_dit_arrival_tests.txt
timeout 1500 python -m pytest tests/test_multigpu_rccl.py -m gpu -q -k "arrival" -s 2>&1 | grep -v "MIOpen(HIP)" | tail -12 | tee gpurun_out/r06_arrival_rank_tests.txt
for layout in auto sp; do
  ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 4 --parallelism $layout --model small --frames 17 --height 128 --width 160 --steps 3 --warmup 1 --no-cpu-baseline \
    2> gpurun_out/r06_bench_4ranks_${layout}_shared.err | tee gpurun_out/r06_bench_selflaunch_4ranks_${layout}_shared_gpu_gloo.json | head -c 400; echo
  tail -25 gpurun_out/r06_bench_4ranks_${layout}_shared.err
done
timeout 1800 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "config2_wan or config3_wan" 2>&1 | grep -v "MIOpen(HIP)" | tail -14 | tee gpurun_out/r06_parity_config23.txt
