#!/bin/bash
# last lease of round 5: smoke() and the driver's bench command at the final HEAD
export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r05_smoke_last.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_bench_last.err | tee gpurun_out/r05_bench_last.json
tail -3 gpurun_out/r05_bench_last.err
