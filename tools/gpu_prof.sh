# rocprofv3 evidence for profiles/: kernel-trace stats of `bench.py` (1 step) and HBM traffic counters of the
# dominant kernel collected in SEPARATE --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass).
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
M=${1:-1.3b}
CMD="python bench.py --model $M --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o s -- $CMD > gpurun_out/prof/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -o w -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/prof/sq -o q -- $CMD > /dev/null 2>&1
python - "$M" <<'PY'
import csv, glob, json, sys, collections, re
m = sys.argv[1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:70]
out = {"model": m, "command": f"python bench.py --model {m} --steps 1 --warmup 1 --no-cpu-baseline (2 steps incl. warm-up)"}
f = glob.glob("gpurun_out/prof/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                        "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:14]]
def pmc(dirn):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"gpurun_out/prof/{dirn}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    return acc, cnt
res = {}
for d in ("fetch", "write", "sq"):
    acc, cnt = pmc(d)
    for k, cs in acc.items():
        if any(t in k for t in ("attn", "gemm", "ln_modulate", "rmsnorm")):
            for c, v in cs.items():
                res.setdefault(k, {})[c] = {"sum": v, "dispatches": cnt[(k, c)], "per_dispatch": v / cnt[(k, c)]}
out["pmc"] = res
# self- vs cross-attention share one kernel and one grid: split the dispatches by their FETCH_SIZE
fs = []
for f in glob.glob("gpurun_out/prof/fetch/**/*counter_collection.csv", recursive=True):
    fs += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
ws = []
for f in glob.glob("gpurun_out/prof/write/**/*counter_collection.csv", recursive=True):
    ws += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
if fs and ws:
    thr = (max(fs) + min(fs)) / 2
    self_f = [x for x in fs if x > thr]
    wthr = (max(ws) + min(ws)) / 2
    self_w = [x for x in ws if x >= wthr] or ws
    fetch_kb, write_kb = sum(self_f) / len(self_f), sum(self_w) / len(self_w)
    out["self_attention_hbm"] = {
        "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb, "launches": len(self_f),
        "fetch_correction": "x2: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)",
        "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
    print("self-attention HBM bytes/launch:", out["self_attention_hbm"]["hbm_bytes_per_launch"] / 1e6, "MB")
json.dump(out, open(f"gpurun_out/prof_summary_{m}.json", "w"), indent=1)
for r in out["kernel_stats"][:8]:
    print(f"{r['pct']:6.2f}%  {r['calls']:5d} x {r['avg_us']:10.1f} us  {r['name']}")
for k, cs in res.items():
    print(k, {c: round(v["per_dispatch"], 1) for c, v in cs.items()})
PY
rm -rf gpurun_out/prof
