# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do
  ICV_DUAL_STREAM=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dual_stream=$v 14b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms')"
done; done
for v in 0 1; do
  ICV_DUAL_STREAM=$v python bench.py --model 1.3b --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dual_stream=$v 1.3b', round(d['value'],4), 'step/s', round(d['ms_per_step'],1), 'ms')"
done
