# round 3, final validation: L2 hit rate of the GEMM, smoke, the whole -m gpu suite with timings, bench lines at HEAD
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gemm_l2_pmc.sh > gpurun_out/gemm_l2.log 2>&1; cat gpurun_out/gemm_l2_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --durations=14 -rf 2>&1 | grep -v "^SKIPPED" | tail -30 | tee gpurun_out/gpu_suite_summary.txt
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_14b_final.json 2> gpurun_out/bench_14b_final.err || tail -5 gpurun_out/bench_14b_final.err
python bench.py --model 1.3b --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_1p3b_final.json 2> gpurun_out/bench_1p3b_final.err || tail -5 gpurun_out/bench_1p3b_final.err
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_i2v720_bf16.json 2> gpurun_out/bench_i2v720_bf16.err || tail -5 gpurun_out/bench_i2v720_bf16.err
python bench.py --model 14b-i2v --frames 93 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/bench_i2v720_fp8.json 2> gpurun_out/bench_i2v720_fp8.err || tail -5 gpurun_out/bench_i2v720_fp8.err
python - <<'PY'
import json
for f in ("bench_14b_final", "bench_1p3b_final", "bench_i2v720_bf16", "bench_i2v720_fp8"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]; c = d["config"]
        print(f, f"{d['value']:.4f} step/s {d['ms_per_step']:.1f} ms/step attn {r['achieved']:.0f} TF frac {r['frac']:.4f} host_enqueue {c['host_enqueue_ms_per_step']:.1f} ms", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
