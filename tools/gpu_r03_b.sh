# round 3, job B: native forward in every mode, guarded RCCL tests (skip reasons), attention variant A/B in ONE process, bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_multigpu_rccl.py -q -m gpu -k "native or sequence_parallel or rccl or bench_line or worker_pool" -rs 2>&1 | tail -25
ATTN_UNIT=1 ATTN_ROUNDS=9 ATTN_VARIANTS=7000,7004,7128,7132 timeout 900 python tools/attn_bench.py "14b self" 2>&1 | tee gpurun_out/attn_pair2.txt
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_14b_r03_a.json 2> gpurun_out/bench_14b_r03_a.err || tail -5 gpurun_out/bench_14b_r03_a.err
python bench.py --steps 3 --warmup 1 --native-forward --no-cpu-baseline > gpurun_out/bench_14b_r03_native.json 2> gpurun_out/bench_14b_r03_native.err || tail -5 gpurun_out/bench_14b_r03_native.err
python - <<'PY'
import json
for f in ("bench_14b_r03_a", "bench_14b_r03_native"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]; c = d["config"]
        print(f, f"{d['value']:.4f} step/s {d['ms_per_step']:.1f} ms/step attn {r['achieved']:.0f} TF frac {r['frac']:.4f} host_enqueue {c['host_enqueue_ms_per_step']:.1f} ms in-loop {c['host_ms_in_timed_loop_per_step']:.1f} ms calls/fwd {c['c_abi_calls_per_forward']}")
    except Exception as e:
        print(f, "FAILED", e)
PY
