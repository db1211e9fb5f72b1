# usage: bash tools/gpu_bench.sh  (on the GPU box via gpurun) — bench + rocprofv3 kernel stats
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --model 1.3b --steps 2 --warmup 1 > gpurun_out/bench_1p3b.json 2> gpurun_out/bench_1p3b.err; tail -2 gpurun_out/bench_1p3b.err; cat gpurun_out/bench_1p3b.json
python bench.py --model 14b --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_14b.json 2> gpurun_out/bench_14b.err; tail -2 gpurun_out/bench_14b.err; cat gpurun_out/bench_14b.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_1p3b -o r01 -- python bench.py --model 1.3b --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_1p3b -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_1p3b -name "*kernel_stats.csv" | head -1); head -25 "$f"
find gpurun_out/prof_1p3b -name "*kernel_trace.csv" -size +20M -delete
