#!/usr/bin/env bash
# round 5, lease D: kernel-level profile of the HIP VAE + first-call wall-clock of a fresh process (2-step calls)
set -uo pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vae -o vae -- python $R/tools/aux_bench.py > $R/gpurun_out/r05d_vae_prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_vae/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/r05d_vae_kernel_stats_top.txt", "w") as out:
    out.write(f"total GPU time {tot/1e6:.1f} ms over 2 encodes + 2 decodes (tools/aux_bench.py)\n")
    for r in rows[:30]:
        out.write(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:120]}\n')
print(open("gpurun_out/r05d_vae_kernel_stats_top.txt").read())
PY
rm -rf gpurun_out/prof_vae
MODEL=14b STEPS=2 FIRST_STEPS=2 timeout 1500 python tools/e2e_wallclock.py 2>&1 | grep -v "MIOpen(HIP)" | tail -8 | tee gpurun_out/r05d_e2e_first_call_2_steps.txt
