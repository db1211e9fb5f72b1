mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_fullsize_gpu.py tests/test_dit_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -s -k "four_step_loop or config1 or sequence_parallel_shards or layer_14b or attention7 or native or fuzz or test_attention" 2>&1 | grep -E "config #|14B block|passed|failed|FAILED|Error|rel-L2|PSNR" | tee gpurun_out/r03_f_tests.txt | tail -40
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/bench_14b_r03_fp8mode.json 2> gpurun_out/bench_fp8.err || tail -3 gpurun_out/bench_fp8.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_14b_r03_b.json 2> gpurun_out/bench_b.err || tail -3 gpurun_out/bench_b.err
python - <<'PY'
import json
for f in ("bench_14b_r03_fp8mode", "bench_14b_r03_b"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]; c = d["config"]
        print(f, f"{d['value']:.4f} step/s {d['ms_per_step']:.1f} ms/step attn {r['achieved']:.0f} TF frac {r['frac']:.4f} host_enqueue {c['host_enqueue_ms_per_step']:.1f} ms")
    except Exception as e:
        print(f, "FAILED", e)
PY
