mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dit_gpu.py -q -s -k config1 2>&1 | grep -E "config #1|passed|failed|Error" | head
