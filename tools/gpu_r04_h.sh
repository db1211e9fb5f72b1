# round 4: suite at HEAD with durations, smoke, the driver's bench command, e2e with the tuned VAE
mkdir -p gpurun_out/r04h; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04h/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=25 -rf -x 2>&1 | grep -v "^SKIPPED" | tail -45 | tee gpurun_out/r04h/gpu_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04h/bench_14b_driver_cmd.json 2> gpurun_out/r04h/bench_14b_driver_cmd.err || tail -5 gpurun_out/r04h/bench_14b_driver_cmd.err
MODEL=14b STEPS=50 python tools/e2e_wallclock.py 2>&1 | tail -1 > gpurun_out/r04h/e2e_generate_14b.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04h/bench_14b_driver_cmd.json")); r = d["roofline"]
print(f"bench {d['value']:.4f} step/s {d['ms_per_step']:.1f} ms attn {r['achieved']:.0f} TF frac {r['frac']:.4f}")
e = json.load(open("gpurun_out/r04h/e2e_generate_14b.json")); print("e2e", e["generate_wallclock_s"], e["non_loop_s"], e["stages_s"])
PY
