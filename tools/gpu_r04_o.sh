mkdir -p gpurun_out/r04o; export TMPDIR=/tmp
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04o/bench_14b_i2v720_bf16.json 2> gpurun_out/r04o/i2v_bf16.err || tail -5 gpurun_out/r04o/i2v_bf16.err
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/r04o/bench_14b_i2v720_fp8_mode.json 2> gpurun_out/r04o/i2v_fp8.err || tail -5 gpurun_out/r04o/i2v_fp8.err
MODEL=1.3b STEPS=50 python tools/e2e_wallclock.py 2>&1 | tail -1 > gpurun_out/r04o/e2e_generate_1p3b.json
GEMM=fp8 MODEL=14b STEPS=50 python tools/e2e_wallclock.py 2>&1 | tail -1 > gpurun_out/r04o/e2e_generate_14b_fp8_mode.json
python - <<'PY'
import json
for f in ("bench_14b_i2v720_bf16", "bench_14b_i2v720_fp8_mode"):
    d = json.load(open(f"gpurun_out/r04o/{f}.json")); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"])
for f in ("e2e_generate_1p3b", "e2e_generate_14b_fp8_mode"):
    d = json.load(open(f"gpurun_out/r04o/{f}.json")); print(f, d["generate_wallclock_s"], d["non_loop_s"])
PY
