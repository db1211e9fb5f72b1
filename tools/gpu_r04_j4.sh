mkdir -p gpurun_out/r04j4; export TMPDIR=/tmp
SCHEDS=3,67 python tools/gemm_sched_bench.py 2>&1 | grep -v amdgpu | head -5 | tee gpurun_out/r04j4/gemm_ksplit_ab_conflict_free.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ksplit" 2>&1 | tail -2
