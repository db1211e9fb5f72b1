mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -s -k "fp8" 2>&1 | grep -E "passed|failed|Error|error|assert|fp8 forward|fp8 loop|differ" | head -30
python tools/gemm_fp8_bench.py 2>&1 | grep -v amdgpu.ids
