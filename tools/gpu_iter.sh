# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
ATTN_UNIT=1 ATTN_VARIANTS=7000,7001,7004,7005 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -3
