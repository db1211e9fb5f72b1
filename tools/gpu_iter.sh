mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_buffers.py -q -m gpu -x 2>&1 | grep -E "passed|failed|outside|Error" | head
python tools/gemm_bench.py 2>&1 | grep -E "qkv|ffn1" | head -4
python tools/ew_bench.py 2>&1 | grep -v amdgpu | head -12
