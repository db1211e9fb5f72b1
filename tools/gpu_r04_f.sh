# (historical: ICV_VAE_NORM=fused here was the stock F.rms_norm variant, measured slower and removed; today ICV_VAE_NORM=hip|stock selects the HIP row kernel)
mkdir -p gpurun_out/r04f; export TMPDIR=/tmp
for pad in copy conv; do for norm in composite fused; do
  ICV_VAE_PAD=$pad ICV_VAE_NORM=$norm python tools/aux_bench.py 2>&1 | grep "^VAE" | tee -a gpurun_out/r04f/vae_layer_tuning.txt
done; done
COMPARE=1 WHAT=decode python tools/aux_bench.py 2>&1 | grep -E "^VAE|vs plain" | tee -a gpurun_out/r04f/vae_layer_tuning.txt
(cd /tmp && WHAT=decode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -o t -- python $GRAFT_REPO_ROOT/tools/aux_bench.py) > gpurun_out/r04f/trace_vae.log 2>&1
f=$(find /tmp/prof_vae -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/r04f/vae_decode_kernel_stats_tuned.csv
timeout 600 python -m pytest tests/test_aux_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r04f/aux_tests.txt
