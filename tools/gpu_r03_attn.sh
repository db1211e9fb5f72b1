# round 3, kernel attempt 1: attn7 128-key publish (variant bit 128) and the short-key launch shape (cross-attention)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention7 or test_attention or attention_add or chunked_state" 2>&1 | tail -3
ATTN_UNIT=1 ATTN_ROUNDS=7 ATTN_VARIANTS=7000,7128 timeout 600 python tools/attn_bench.py "14b self" "1.3b self" "sp4 14b" 2>&1 | tee gpurun_out/attn_pair.txt
ATTN_UNIT=1 ATTN_ROUNDS=5 ATTN_VARIANTS=7004,7132 timeout 600 python tools/attn_bench.py "14b self" 2>&1 | tee -a gpurun_out/attn_pair.txt
ATTN_UNIT=1 ATTN_ROUNDS=7 ATTN_ITERS=20 ATTN_VARIANTS=7000,8000 timeout 600 python tools/attn_bench.py "14b cross" "14b ximg" "sp4 cross" 2>&1 | tee gpurun_out/attn_short.txt
echo "--- self-launch, 2 ranks sharing the GPU over gloo (code-path check only)"
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --model small --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/selflaunch.err | tee gpurun_out/selflaunch.json | cut -c1-600
tail -3 gpurun_out/selflaunch.err
ICV_BENCH_SHARE_GPU=1 ICV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --model small --steps 2 --warmup 1 --no-cpu-baseline --parallelism sp 2> gpurun_out/selflaunch4.err | tee gpurun_out/selflaunch4.json | cut -c1-600
tail -3 gpurun_out/selflaunch4.err
