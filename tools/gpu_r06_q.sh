#!/usr/bin/env bash
# round 6 (q): attn7p with the trimmed scalar work in its LDS-DMA issue: bit-identity + pieces tests, timing vs attn7.hip, the driver's bench command
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attn_pieces_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "pieces or attention" 2>&1 | grep -v "MIOpen(HIP)" | tail -5 | tee gpurun_out/r06_attn7p_trim_tests.txt
timeout 300 python tools/attn7p_vs_attn7.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tail -8 | tee gpurun_out/r06_attn7p_trim_vs_attn7.txt
FUZZ_R6=1 timeout 300 python tools/fuzz_kernels.py 90 7 2>&1 | tail -3 | tee gpurun_out/r06_attn7p_trim_fuzz.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_bench_14b.err | tee gpurun_out/r06_bench_14b_attn7p_trim.json | cut -c1-400
