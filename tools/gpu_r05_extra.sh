#!/usr/bin/env bash
# round 5, extra records at the final HEAD: config #5's workload on one GPU (bf16 and the e4m3 mode: attn8.hip changed this round), the
# 1.3B profile, the other e2e wall-clocks (first call AND steady state)
set -uo pipefail
mkdir -p gpurun_out/r05x; export TMPDIR=/tmp
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05x/bench_14b_i2v720_bf16.json 2> gpurun_out/r05x/i2v_bf16.err || tail -5 gpurun_out/r05x/i2v_bf16.err
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/r05x/bench_14b_i2v720_fp8_mode.json 2> gpurun_out/r05x/i2v_fp8.err || tail -5 gpurun_out/r05x/i2v_fp8.err
MODEL=1.3b STEPS=50 timeout 900 python tools/e2e_wallclock.py 2>&1 | tail -1 > gpurun_out/r05x/e2e_generate_1p3b.json
GEMM=fp8 MODEL=14b STEPS=50 timeout 1200 python tools/e2e_wallclock.py 2>&1 | tail -1 > gpurun_out/r05x/e2e_generate_14b_fp8_mode.json
timeout 900 bash tools/gpu_prof_r05.sh 1.3b > gpurun_out/r05x/prof_1p3b.log 2>&1
python - <<'PY'
import json
for f in ("bench_14b_i2v720_bf16", "bench_14b_i2v720_fp8_mode"):
    d = json.loads(open(f"gpurun_out/r05x/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"])
for f in ("e2e_generate_1p3b", "e2e_generate_14b_fp8_mode"):
    d = json.load(open(f"gpurun_out/r05x/{f}.json")); print(f, d["first_call_s"], d["generate_wallclock_s"], d["non_loop_s"])
PY
head -8 gpurun_out/prof_r05_1.3b.md
