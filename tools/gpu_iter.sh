mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k gemm 2>&1 | tail -2
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
