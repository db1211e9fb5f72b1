#!/usr/bin/env bash
# Effective clock and issue/stall breakdown of the gemm256 variants on ONE DiT shape (default 14B FFN1).
#   effective clock = GRBM_GUI_ACTIVE / kernel duration;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * GRBM_GUI_ACTIVE)
# Counters are collected in their own rocprofv3 passes (--kernel-trace + --pmc only), never with other trace domains.
export TMPDIR=/tmp
out=gpurun_out/gemm_pmc; rm -rf $out; mkdir -p $out
SHAPE=${1:-"37440,13824,5120,1"}
cat > /tmp/gemm_one.py <<PY
import sys, os, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from infinicube_amd.videogen.ops import HipOps
M, N, K, epi = [int(x) for x in "$SHAPE".split(",")]
ops = HipOps("cuda:0")
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn((N,), device="cuda")
out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
kw = dict(resid=out, gate=bias) if epi == 2 else {}
ops.lib.icv_set_option(b"gemm256", 1)
for sched, mf in ((0, 16), (1, 16), (3, 16), (0, 32)):
    ops.lib.icv_set_option(b"gemm256_sched", sched); ops.lib.icv_set_option(b"gemm256_mfma", mf)
    for _ in range(6):
        ops.gemm(a, w, bias, out, epi, **kw)
    torch.cuda.synchronize()
for _ in range(6):
    torch.nn.functional.linear(a, w)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python /tmp/gemm_one.py > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/sq -o q -- python /tmp/gemm_one.py > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/sq2 -o q -- python /tmp/gemm_one.py > $out/sq2.log 2>&1
python - <<'PY'
import csv, glob, collections, re
def short(n): return re.sub(r"\(anonymous namespace\)::", "", n)[:80]
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/gemm_pmc/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq", "sq2"):
    for f in glob.glob(f"gpurun_out/gemm_pmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/gemm_pmc_summary.txt", "w") as o:
    for k, ds in dur.items():
        if not any(t in k for t in ("gemm", "Cijk", "gemm256")):
            continue
        ds = sorted(ds)[: max(1, len(ds) - 2)]          # drop the slowest two (warm-up)
        d = sum(ds) / len(ds)
        c = {n: sum(v) / len(v) for n, v in acc.get(k, {}).items()}
        line = f"{k}: avg {d / 1e3:.1f} us"
        if "GRBM_GUI_ACTIVE" in c:
            # NOTE: the PMC pass runs at its own (lower) clock; the clock below is that pass's, from its own durations
            line += f" | GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.3g}"
        for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA",
                  "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"):
            if n in c:
                line += f" | {n} {c[n]:.4g}"
        o.write(line + "\n"); print(line)
    # durations inside the PMC pass, for the clock
    for d in ("sq",):
        for f in glob.glob(f"gpurun_out/gemm_pmc/{d}/**/*kernel_trace.csv", recursive=True):
            dd = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                dd[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k, v in dd.items():
                if any(t in k for t in ("gemm", "Cijk")):
                    v = sorted(v)[: max(1, len(v) - 2)]
                    g = acc.get(k, {}).get("GRBM_GUI_ACTIVE")
                    if g:
                        line = f"[pmc pass] {k}: avg {sum(v) / len(v) / 1e3:.1f} us -> clock {sum(g) / len(g) / (sum(v) / len(v)):.3f} GHz"
                        o.write(line + "\n"); print(line)
PY
