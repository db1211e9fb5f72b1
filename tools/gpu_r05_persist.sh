#!/bin/bash
# the persistent GEMM as the default for the bf16 / GELU epilogues: tests, kernel A/B, whole-step A/B on ONE box (interleaved)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -4 | tee gpurun_out/r05_persist_tests.txt
timeout 600 python tools/gemm_persistent_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_persist_ab_kernels.txt
for rep in 1 2; do
  for p in 0 1; do
    ICV_OPTIONS="gemm256_persist=$p" timeout 400 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gemm256_persist=$p rep $rep: %.1f ms/step, %.4f steps/s, attention %.0f TF/s' % (d['ms_per_step'], d['value'], d['roofline']['achieved']))" | tee -a gpurun_out/r05_persist_ab_step.txt
  done
done
