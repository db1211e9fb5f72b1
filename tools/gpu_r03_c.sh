# round 3, job C: the whole -m gpu suite at HEAD (timings), attention variant 132 on the other shapes
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --durations=12 -rs 2>&1 | grep -v "^SKIPPED.*needs >= " | tail -40 | tee gpurun_out/gpu_suite_summary.txt
ATTN_UNIT=1 ATTN_ROUNDS=7 ATTN_VARIANTS=7000,7132 timeout 900 python tools/attn_bench.py "1.3b self" "sp4 14b" "sp8 14b" 2>&1 | tee gpurun_out/attn_pair3.txt
