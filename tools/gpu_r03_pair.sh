mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_gpu.py -m gpu -q -rf -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for b in 0 1 0 1; do ICV_CFG_BATCH=$b python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_pair_$b.json; python - <<PY
import json
d=json.load(open("gpurun_out/bench_pair_$b.json")); print("ICV_CFG_BATCH=$b", round(d["ms_per_step"],1), "ms/step", round(d["value"],4), "step/s attn", round(d["roofline"]["achieved"]), "calls/fwd", d["config"]["c_abi_calls_per_forward"])
PY
done
for b in 0 1; do ICV_CFG_BATCH=$b python bench.py --model 1.3b --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.3b ICV_CFG_BATCH=$b', round(d['ms_per_step'],1))"; done
