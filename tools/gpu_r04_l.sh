# round 4: config #3 (Wan2.1-14B, S = 37 440) at the stated 50 steps at HEAD, bf16 arm (a GPU lease is one hour: ~49 min of fp32 oracle)
mkdir -p gpurun_out/r04l; export TMPDIR=/tmp
ICV_SLOW_TESTS=2 ICV_SLOW_ARMS=bf16 timeout 3500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "test_configs_2_and_3_at_50_steps and 14b" 2>&1 | grep -v "^SKIPPED" | tail -6 | tee gpurun_out/r04l/parity_config3_50_steps_log.txt
cp gpurun_out/parity_config3_50_steps.txt gpurun_out/r04l/ 2>/dev/null
