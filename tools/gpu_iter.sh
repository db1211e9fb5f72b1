mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dit_gpu.py -q -s -k "config1" 2>&1 | grep -E "passed|failed|Error|error|assert|config #1" | head -30
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_fp8 -o fp8 -- python /root/repo/bench.py --gemm-dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_fp8_bench.log 2>&1
f=$(find /root/repo/gpurun_out/prof_fp8 -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-220; find /root/repo/gpurun_out/prof_fp8 -name "*kernel_trace.csv" -delete
