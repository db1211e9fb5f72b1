mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -k "attention_add or i2v or rccl or sequence_parallel" 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
python bench.py --model 1.3b --frames 17 --height 256 --width 448 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_cfg1.json; cat gpurun_out/bench_cfg1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1', d['value'], d['ms_per_step'], d['config']['frac_of_bf16_mfma_peak'])"
