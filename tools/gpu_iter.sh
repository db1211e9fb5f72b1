mkdir -p gpurun_out; export TMPDIR=/tmp
ATTN_UNIT=1 ATTN_VARIANTS=1012,7004,7005 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
for v in "attn_kernel=2" "attn_kernel=7,attn7_variant=5" "attn_kernel=7,attn7_variant=4"; do
  ICV_OPTIONS="$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms  attn', round(d['roofline']['achieved']), 'TF')"
done; done
