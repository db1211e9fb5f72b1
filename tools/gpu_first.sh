set -x
mkdir -p gpurun_out
./tools/probe_tr > gpurun_out/probe_tr.txt 2>&1; tail -3 gpurun_out/probe_tr.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40
