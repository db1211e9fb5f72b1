#!/bin/bash
# last lease of round 5: smoke() and the kernel tests on the library as committed
export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r05_smoke_last.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -m gpu -q 2>&1 | tail -3 | tee -a gpurun_out/r05_smoke_last.txt
