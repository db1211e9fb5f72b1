mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_buffers.py -q -m gpu 2>&1 | tail -4
