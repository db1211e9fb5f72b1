# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_gpu.py -q -s -k "config1" 2>&1 | grep -E "passed|failed|Error|assert|config #1" | head
for a in "bf16 bf16" "fp8 bf16" "fp8 fp8"; do set -- $a
  python bench.py --gemm-dtype $1 --attn-dtype $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gemm $1 attn $2 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done
