# (historical: this job A/B-ed ICV_VAE_CONV=split2d, a variant that measured slower and was removed again - profiles/r04/vae_layer_tuning.md)
mkdir -p gpurun_out/r04e; export TMPDIR=/tmp
for mode in native split2d; do
  ICV_VAE_CONV=$mode python tools/aux_bench.py 2>&1 | grep "^VAE" | tee -a gpurun_out/r04e/vae_conv_modes.txt
done
ICV_VAE_CONV=split2d COMPARE=1 WHAT=decode python tools/aux_bench.py 2>&1 | grep -E "^VAE|vs native" | tee -a gpurun_out/r04e/vae_conv_modes.txt
for mode in native split2d; do
  (cd /tmp && ICV_VAE_CONV=$mode WHAT=decode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae_$mode -o t -- python $GRAFT_REPO_ROOT/tools/aux_bench.py) > gpurun_out/r04e/trace_vae_$mode.log 2>&1
  f=$(find /tmp/prof_vae_$mode -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/r04e/vae_decode_kernel_stats_$mode.csv
done
MODEL=14b STEPS=50 python tools/e2e_wallclock.py 2>&1 | tail -4 | tee gpurun_out/r04e/e2e_generate_14b.txt
