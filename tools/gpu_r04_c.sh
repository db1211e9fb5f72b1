# round 4: shard-shape attention study, compute-only projection with the paired sp forward, the changed tests
mkdir -p gpurun_out/r04c; export TMPDIR=/tmp
python tools/sp_attn_shapes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c/sp_attn_shapes.txt
python tools/sp_shard_compute_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c/sp_projection.txt
cp gpurun_out/sp_compute_only_projection.json gpurun_out/r04c/
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_dit_gpu.py -m gpu -q -x --durations=8 -k "config3 or config2 or sequence_parallel_shards or cfg_batched or native_forward" -s 2>&1 | grep -v "^SKIPPED" | tail -40 | tee gpurun_out/r04c/tests.txt
