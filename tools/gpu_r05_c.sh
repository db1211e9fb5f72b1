#!/usr/bin/env bash
# round 5, lease C: e4m3 on the wire (kernel + DiT + processes sharing the GPU), then a kernel-level profile of the HIP VAE
set -uo pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "fp8" -q 2>&1 | tail -15 | tee gpurun_out/r05c_fp8_kernels.txt
timeout 900 python -m pytest tests/test_dit_gpu.py -k "sequence_parallel or fp8 or copy_engine or native_forward" -q 2>&1 | tail -15 | tee gpurun_out/r05c_dit.txt
timeout 1200 python -m pytest tests/test_multigpu_rccl.py -k "shared_gpu_fp8" -q -s 2>&1 | tail -15 | tee gpurun_out/r05c_shared_fp8.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vae -o vae -- python $GRAFT_REPO_ROOT/tools/aux_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r05c_vae_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_vae/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/r05c_vae_kernel_stats_top.txt", "w") as out:
    for r in rows[:25]:
        out.write(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}\n')
print(open("gpurun_out/r05c_vae_kernel_stats_top.txt").read())
PY
rm -rf gpurun_out/prof_vae
