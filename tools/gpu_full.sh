# full GPU check: smoke + all -m gpu tests + both bench configs; results under gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
python bench.py --model 1.3b --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1p3b.json 2> gpurun_out/bench_1p3b.err || tail -5 gpurun_out/bench_1p3b.err
python bench.py > gpurun_out/bench_14b.json 2> gpurun_out/bench_14b.err || tail -5 gpurun_out/bench_14b.err
python - <<'PY'
import json
for m in ("1p3b", "14b"):
    try:
        d = json.load(open(f"gpurun_out/bench_{m}.json"))
        r = d["roofline"]
        print(m, f"{d['value']:.4f} step/s  {d['ms_per_step']:.1f} ms/step  model {d['config']['model_tflops_all_gpus']:.0f} TF ({100*d['config']['frac_of_bf16_mfma_peak']:.1f}%)  attn {r['achieved']:.0f} TF ({r['avg_launch_ms']:.2f} ms) traffic {r['traffic']}", d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(m, "FAILED", e)
PY
