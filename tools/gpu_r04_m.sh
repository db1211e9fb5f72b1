mkdir -p gpurun_out/r04m; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multigpu_rccl.py tests/test_vae_shard.py tests/test_dit_gpu.py tests/test_aux_gpu.py -m gpu -q --durations=6 -k "tiled_vae or sharing_it or native_forward_matches or hip_norm or rmsnorm_act or falls_back or one_json" 2>&1 | grep -v "^SKIPPED" | tail -14 | tee gpurun_out/r04m/new_tests.txt
python bench.py --model 1.3b --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04m/bench_1p3b.json 2> gpurun_out/r04m/bench_1p3b.err || tail -5 gpurun_out/r04m/bench_1p3b.err
python bench.py --gemm-dtype fp8 --attn-dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04m/bench_14b_fp8_mode.json 2> gpurun_out/r04m/bench_14b_fp8_mode.err || tail -5 gpurun_out/r04m/bench_14b_fp8_mode.err
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo python bench.py --gpus 4 --model small --frames 17 --height 256 --width 448 --steps 2 --warmup 1 > gpurun_out/r04m/bench_selflaunch_4ranks_shared_gpu_gloo.json 2> gpurun_out/r04m/bench_selflaunch_4ranks.err || tail -5 gpurun_out/r04m/bench_selflaunch_4ranks.err
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo ICV_GUARD_INJECT=0:2:groups:raise python bench.py --gpus 4 --model small --frames 17 --height 256 --width 448 --steps 2 --warmup 1 > gpurun_out/r04m/bench_selflaunch_4ranks_injected_failure.json 2> /dev/null
python - <<'PY'
import json
for f in ("bench_1p3b", "bench_14b_fp8_mode", "bench_selflaunch_4ranks_shared_gpu_gloo", "bench_selflaunch_4ranks_injected_failure"):
    try:
        d = json.load(open(f"gpurun_out/r04m/{f}.json")); r = d["roofline"]
        print(f, f"{d['value']:.4f} step/s {d['ms_per_step']:.1f} ms attn {r['achieved']:.0f} TF", (d.get("multi_gpu") or {}).get("plan"), len((d.get("multi_gpu") or {}).get("failed_attempts") or []),
              ((d.get("multi_gpu") or {}).get("autotune") or {}).get("chosen"))
    except Exception as e:
        print(f, "FAILED", e)
PY
