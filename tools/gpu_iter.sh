set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -15
timeout 300 python tools/gemm_bench.py 2>&1 | tail -12
