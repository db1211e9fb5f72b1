mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "4wave" 2>&1 | grep -E "passed|failed|outside|Error|max err" | head
python tools/gemm_bench.py 2>&1 | grep -v amdgpu | awk -F'|' '{print $1 "|" $2 "|" $5}' | cut -c1-140
