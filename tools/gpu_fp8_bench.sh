# fp8 / i2v bench lines (BASELINE.json config #5 workloads on ONE GPU); results under gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --gemm-dtype fp8 --no-cpu-baseline > gpurun_out/bench_14b_fp8.json 2> gpurun_out/b1.err || tail -5 gpurun_out/b1.err
python bench.py --gemm-dtype fp8 --attn-dtype fp8 --no-cpu-baseline > gpurun_out/bench_14b_fp8_attn8.json 2> gpurun_out/b4.err || tail -5 gpurun_out/b4.err
python bench.py --model 14b-i2v --height 720 --width 1280 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_14b_i2v_720p_bf16.json 2> gpurun_out/b2.err || tail -5 gpurun_out/b2.err
python bench.py --model 14b-i2v --height 720 --width 1280 --steps 1 --warmup 1 --gemm-dtype fp8 --attn-dtype fp8 --no-cpu-baseline > gpurun_out/bench_14b_i2v_720p_fp8_attn8.json 2> gpurun_out/b3.err || tail -5 gpurun_out/b3.err
python - <<'PY'
import json
for m in ("14b_fp8", "14b_fp8_attn8", "14b_i2v_720p_bf16", "14b_i2v_720p_fp8_attn8"):
    try:
        d = json.load(open(f"gpurun_out/bench_{m}.json"))
        r = d["roofline"]
        print(m, f"{d['value']:.4f} step/s  {d['ms_per_step']:.1f} ms/step  model {d['config']['model_tflops_all_gpus']:.0f} TF  attn {r['achieved']:.0f} TF ({r['avg_launch_ms']:.2f} ms)")
    except Exception as e:
        print(m, "FAILED", e)
PY
