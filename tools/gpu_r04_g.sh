mkdir -p gpurun_out/r04g; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_aux_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r04g/aux_tests.txt
for norm in stock hip; do
  ICV_VAE_NORM=$norm python tools/aux_bench.py 2>&1 | grep "^VAE" | sed "s/^/norm=$norm /" | tee -a gpurun_out/r04g/vae_hip_norm.txt
done
(cd /tmp && WHAT=decode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -o t -- python $GRAFT_REPO_ROOT/tools/aux_bench.py) > gpurun_out/r04g/trace_vae.log 2>&1
f=$(find /tmp/prof_vae -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/r04g/vae_decode_kernel_stats_tuned.csv
