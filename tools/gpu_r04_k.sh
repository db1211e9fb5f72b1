# round 4, final validation at HEAD: smoke, the whole -m gpu suite, fuzz, the driver's bench command, rocprofv3 stats + PMC passes
mkdir -p gpurun_out/r04k; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04k/smoke.txt
timeout 1300 python -m pytest tests -m gpu -q --durations=12 -rf 2>&1 | grep -v "^SKIPPED" | tail -30 | tee gpurun_out/r04k/gpu_suite.txt
timeout 300 python tools/fuzz_kernels.py 90 4 2>&1 | tail -4 | tee gpurun_out/r04k/fuzz_kernels.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04k/bench_14b_final.json 2> gpurun_out/r04k/bench_14b_final.err || tail -5 gpurun_out/r04k/bench_14b_final.err
bash tools/gpu_prof_r04.sh 14b > gpurun_out/r04k/prof.log 2>&1; tail -18 gpurun_out/r04k/prof.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04k/bench_14b_final.json")); r = d["roofline"]
print(f"bench {d['value']:.4f} step/s {d['ms_per_step']:.1f} ms attn {r['achieved']:.0f} TF frac {r['frac']:.4f}")
PY
