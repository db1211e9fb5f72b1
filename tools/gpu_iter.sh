# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --steps 2 --warmup 1 --no-cpu-baseline | tail -1
