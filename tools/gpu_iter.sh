mkdir -p gpurun_out; export TMPDIR=/tmp
export ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo
for cfg in "2 auto" "4 auto" "2 sp" "4 sp"; do
  set -- $cfg
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2951$1 bench.py --gpus $1 --parallelism $2 --model small --frames 17 --height 128 --width 160 --steps 2 --warmup 1 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\|OMP_NUM" | tail -3 | cut -c1-700
done
unset ICV_BENCH_SHARE_GPU ICV_DIST_BACKEND
python bench.py --model small --frames 17 --height 128 --width 160 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
