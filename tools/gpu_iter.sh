mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "ln_ or norm" 2>&1 | tail -2
python tools/ew_bench.py 2>&1 | grep -v amdgpu
