mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" 2>&1 | grep -E "passed|failed|outside|Error" | head
python tools/gemm_bench.py 2>&1 | cut -c1-110
