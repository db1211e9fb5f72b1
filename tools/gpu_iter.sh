mkdir -p gpurun_out; export TMPDIR=/tmp
ATTN_VARIANTS=5,37,69,133,229,253 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 14b 2>&1 | grep -v amdgpu.ids
