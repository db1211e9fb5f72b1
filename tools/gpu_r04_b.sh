# round 4: the whole -m gpu suite at HEAD with per-test durations (suite-time budget: <= 600 s on one GPU)
mkdir -p gpurun_out/r04b; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1700 python -m pytest tests -m gpu -q --durations=40 -rf -x 2>&1 | grep -v "^SKIPPED" | tail -70 | tee gpurun_out/r04b/gpu_suite.txt
