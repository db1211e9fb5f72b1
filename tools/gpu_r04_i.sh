# round 4: configs #2 / #3 at the stated 50 steps at HEAD (recorded opt-in runs; ~60 GPU-minutes of fp32 oracle)
mkdir -p gpurun_out/r04i; export TMPDIR=/tmp
ICV_SLOW_TESTS=2 timeout 5400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "test_configs_2_and_3_at_50_steps or test_vae" 2>&1 | grep -v "^SKIPPED" | tail -12 | tee gpurun_out/r04i/parity_50_steps.txt
cp gpurun_out/parity_config2_50_steps.txt gpurun_out/parity_config3_50_steps.txt gpurun_out/r04i/ 2>/dev/null
