mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "minimum or ragged or single" 2>&1 | tail -8
