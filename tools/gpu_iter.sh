# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_gpu.py -q -s 2>&1 | grep -E "passed|failed|Error|error|assert|8 steps|config #1" | head -20
python bench.py --model 1.3b --frames 17 --height 256 --width 448 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 eager (bench)', d['value'], d['ms_per_step'])"
