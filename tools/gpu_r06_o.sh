#!/usr/bin/env bash
# round 6: the software-pipelined attn8 loop as default - every fp8-mode test + the fuzzer + the A/B record
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
STAGE=${1:-all}
timeout 250 python tools/attn8_pipe_debug.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tail -8
ATTN8_VARIANTS=0,8,32,160,164,0,164 timeout 500 python tools/attn_fp8_bench.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee gpurun_out/r06_attn8_pipe_ab.txt | cut -c1-700
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_multigpu_rccl.py -m gpu -q -k "fp8 or e4m3 or config1" 2>&1 | grep -v "MIOpen(HIP)" | tail -8 | tee gpurun_out/r06_attn8_pipe_tests.txt
FUZZ_FP8=1 timeout 600 python tools/fuzz_kernels.py 120 6 2>&1 | tail -6 | tee gpurun_out/r06_attn8_pipe_fuzz.txt
if [[ $STAGE == all ]]; then
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "i2v_720p or shards_full_S or config3" 2>&1 | grep -v "MIOpen(HIP)" | tail -14 | tee -a gpurun_out/r06_attn8_pipe_tests.txt
fi
if [[ $STAGE == all ]]; then
timeout 600 python bench.py --gemm-dtype fp8 --attn-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_bench_14b_fp8_mode_pipelined.json
timeout 900 python bench.py --model 14b-i2v --height 720 --width 1280 --gemm-dtype fp8 --attn-dtype fp8 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06_bench_14b_i2v720_fp8_mode_pipelined.json
fi
