# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -k "attention_fp8 or sequence_parallel_path" 2>&1 | grep -E "passed|failed|rms err|Error|rel-L2|assert" | head -10
