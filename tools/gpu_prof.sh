# rocprofv3 kernel stats for one bench step per model; copies the small summaries to gpurun_out/
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in 1.3b 14b; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o r01 -- python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench_$m.log 2>&1
  f=$(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1); echo "== $m $f"; head -20 "$f"
  cp "$f" gpurun_out/kernel_stats_$m.csv
  rm -rf gpurun_out/prof_$m
done
