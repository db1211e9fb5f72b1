# scratch: one-off GPU experiment of the moment (run with: gpurun -- 'bash tools/gpu_iter.sh')
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 4 5; do echo "attn8_variant=$v"; ICV_OPTIONS="attn8_variant=$v" python tools/attn_fp8_bench.py 2>&1 | grep -E "14b self|sp4" | cut -c1-150; done
