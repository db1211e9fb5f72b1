#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_so -o so -- python $R/tools/probe_streamops.py 2>&1 | grep -v "^W2026\|^E2026" | tail -6
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_so/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(f'  {r["Name"][:90]:90s} calls {r["Calls"]:>4} avg {float(r["AverageNs"])/1e3:10.1f} us max {float(r["MaxNs"])/1e3:10.1f} us')
PY
rm -rf gpurun_out/prof_so
