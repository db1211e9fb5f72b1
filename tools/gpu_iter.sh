mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | grep -E "passed|failed|outside|rms err|Error|determin" | head
ATTN_VARIANTS=1004,6005 ATTN_ROUNDS=3 ATTN_ITERS=2 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
