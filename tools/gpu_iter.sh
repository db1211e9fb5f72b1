mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "unit_scale" 2>&1 | grep -E "passed|failed|outside|rms err|Error" | head -5
ICV_OPTIONS="attn_cinit=1" timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "unit_scale" 2>&1 | grep -E "passed|failed|outside|rms err|Error" | head -5
for rep in 1 2; do
for v in 0 1; do
  ICV_OPTIONS="attn_cinit=$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cinit=$v 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done; done
