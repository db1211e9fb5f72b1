mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -3
ATTN_VARIANTS=1004,4005 ATTN_ROUNDS=2 ATTN_ITERS=2 python tools/attn_bench.py 14b 2>&1 | grep -v amdgpu.ids | head -2
