mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in 4 12; do
  ICV_OPTIONS="attn2_variant=$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('attn2_variant=$v 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done; done
for v in 4 12; do
  ICV_OPTIONS="attn2_variant=$v" python bench.py --model 1.3b --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('attn2_variant=$v 1.3b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done
