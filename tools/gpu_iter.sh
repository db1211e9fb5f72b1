mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dit_gpu.py -q -k "generator_end or forward_parity" 2>&1 | tail -2
