#!/bin/bash
# whole-step A/B of the persistent GEMM default at Wan2.1-1.3B (same box, interleaved)
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/r05_persist_ab_step_1p3b.txt
for rep in 1 2 3; do
  for p in 0 1; do
    ICV_OPTIONS="gemm256_persist=$p" timeout 300 python bench.py --gpus 1 --model 1.3b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1.3B gemm256_persist=$p rep $rep: %.1f ms/step, %.4f steps/s, attention %.0f TF/s' % (d['ms_per_step'], d['value'], d['roofline']['achieved']))" | tee -a gpurun_out/r05_persist_ab_step_1p3b.txt
  done
done
