mkdir -p gpurun_out/r04j; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ksplit or fuzz" 2>&1 | tail -4 | tee gpurun_out/r04j/ksplit_tests.txt
SCHEDS=3,67 python tools/gemm_sched_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04j/gemm_ksplit_ab.txt
