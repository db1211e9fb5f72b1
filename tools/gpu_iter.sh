mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or gelu" 2>&1 | grep -E "passed|failed|outside|Error" | head
python tools/gemm_bench.py 2>&1 | grep -E "ffn1"
