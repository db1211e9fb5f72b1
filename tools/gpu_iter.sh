mkdir -p gpurun_out; export TMPDIR=/tmp
ATTN_ZERO=1 ATTN_VARIANTS=1004,4005,253 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 14b 2>&1 | grep -v amdgpu.ids
ATTN_VARIANTS=1004,4005,253 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 14b 2>&1 | grep -v amdgpu.ids
