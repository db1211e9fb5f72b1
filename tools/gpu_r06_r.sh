#!/usr/bin/env bash
# round 6 (r): counters of attn7q (software-pipelined bf16 attention) next to attn7p in one process
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
D=gpurun_out/prof_r06_attn7q; rm -rf $D; mkdir -p $D
ATTN7Q_TIMING_ONLY=1 timeout 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $D -o q -- python tools/attn7q_ab.py > $D/run.log 2>&1
python - $D <<'PY' | tee gpurun_out/r06_attn7q_counters.txt
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn7" not in n: continue
        key = (n[:48], r.get("Grid_Size") or r.get("Grid_Size_X"))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, c in rows.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    dur = a["dur_us"]
    clk = a["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3 if "GRBM_GUI_ACTIVE" in a else 0
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (clk * 1e3 * dur) if clk else 0
    wc = a.get("SQ_WAVE_CYCLES", 1)
    print(f"{key[0]:50s} grid {key[1]:>9s} launches {len(c['dur_us'])//1:4d} avg {dur:9.1f} us  eff clock {clk:.2f} GHz  MFMA busy {busy:.3f}  wait/stall/active {a.get('SQ_WAIT_ANY',0)/wc:.2f}/{a.get('SQ_WAIT_INST_ANY',0)/wc:.2f}/{a.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}  VALU {a.get('SQ_INSTS_VALU',0):.3g} SALU {a.get('SQ_INSTS_SALU',0):.3g}")
PY
rm -rf $D
