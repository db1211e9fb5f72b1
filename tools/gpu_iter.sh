mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -k "attention or forward or loop or sequence or i2v" 2>&1 | grep -E "passed|failed|outside|rms err|Error|determin|assert" | head
for rep in 1 2; do
for v in 0 1; do
  ICV_OPTIONS="attn_unit_scale=$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unit_scale=$v 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done; done
