#!/usr/bin/env bash
# Round-6 profile of ANY `bench.py` configuration (one timed denoise step + one warm-up step), generalised from tools/gpu_prof_r05.sh:
#   pass 1  rocprofv3 --kernel-trace --stats                      -> per-kernel time
#   pass 2  --pmc GRBM_GUI_ACTIVE + SQ wave / wait / MFMA          -> effective clock and MFMA-busy AT THAT CLOCK per kernel group
#   pass 3  --pmc SQ VALU / LDS instruction + LDS bank-conflict    -> VALU and LDS instruction mix, LDS conflict share
#   pass 4  --pmc SQ_ACTIVE_INST_{VALU,LDS,..} (if the names exist on this ROCm; the pass is allowed to fail) -> VALU-busy / LDS-busy share of wave cycles
#   pass 5/6 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)       -> HBM bytes per launch (FETCH_SIZE x2: gfx950 correction)
# Counter passes use --kernel-trace + --pmc only (never with other trace domains).
# Usage: tools/gpu_prof_r06.sh TAG S -- <bench.py arguments>     -> gpurun_out/prof_r06_TAG.{json,md} (+ _kernel_stats.csv, _bench_line.json)
export TMPDIR=/tmp
TAG=$1; S=$2; shift 3
D=gpurun_out/prof_r06_$TAG; rm -rf $D; mkdir -p $D
CMD="python bench.py $* --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o s -- $CMD > $D/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $D/sq -o q -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $D/sq2 -o q -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $D/sq3 -o q -- $CMD > $D/sq3.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o w -- $CMD > /dev/null 2>&1
python - "$TAG" "$D" "$S" "$CMD" <<'PY'
import collections, csv, glob, json, re, sys
tag, D, S, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
d = 5120
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:72]
def keep(n): return any(t in n for t in ("attn", "gemm", "ln_modulate", "rmsnorm", "patchify", "unpatchify", "quant", "fp8", "absmax"))
def label(name, grid, dur_us):
    if "attn" in name:
        return "self-attention (K6)" if dur_us > 3000 else "cross-attention (K9)"
    if "gemm" in name:
        return f"grid {grid}"
    return ""
def groups(path_glob, with_counters=False):
    out = collections.defaultdict(lambda: {"dur": [], "ctr": collections.defaultdict(list)})
    for f in glob.glob(path_glob, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if not keep(n): continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
            key = (n, label(n, grid, dur))
            if with_counters:
                out[key]["ctr"][r["Counter_Name"]].append(float(r["Counter_Value"]))
                did = (r["Dispatch_Id"], key)
                if did in seen: continue
                seen.add(did)
            out[key]["dur"].append(dur)
    return out
st = groups(f"{D}/stats/**/*kernel_trace.csv")
sq = groups(f"{D}/sq/**/*counter_collection.csv", True)
s2 = groups(f"{D}/sq2/**/*counter_collection.csv", True)
s3 = groups(f"{D}/sq3/**/*counter_collection.csv", True)
fe = groups(f"{D}/fetch/**/*counter_collection.csv", True)
wr = groups(f"{D}/write/**/*counter_collection.csv", True)
tot = sum(sum(v["dur"]) for v in st.values())
avg = lambda v: sum(v) / len(v) if v else None
def match(tbl, key):
    """the self / cross split is by duration, and counter passes run slower: fall back to the same kernel name + label"""
    return tbl.get(key)
rows = []
for key, v in sorted(st.items(), key=lambda kv: -sum(kv[1]["dur"])):
    r = {"kernel": key[0], "what": key[1], "calls": len(v["dur"]), "avg_us": avg(v["dur"]), "pct_of_gpu_time": 100 * sum(v["dur"]) / tot}
    q = match(sq, key)
    if q and q["ctr"].get("GRBM_GUI_ACTIVE"):
        gui = avg(q["ctr"]["GRBM_GUI_ACTIVE"]) / 8.0          # the counter sums the 8 XCDs' GRBMs
        r["pmc_pass_avg_us"] = avg(q["dur"])
        r["effective_clock_ghz"] = gui / (avg(q["dur"]) * 1e3)
        mf = avg(q["ctr"].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
        r["mfma_busy_frac_at_effective_clock"] = mf / (1024.0 * gui) if gui else None
        wc = avg(q["ctr"].get("SQ_WAVE_CYCLES", [0]))
        if wc:
            r["wave_cycles_split"] = {k: avg(q["ctr"].get(n, [0])) / wc for k, n in
                                      (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"))}
        r["insts_mfma_per_launch"] = avg(q["ctr"].get("SQ_INSTS_MFMA", [0]))
    x = match(s2, key)
    if x:
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"):
            if x["ctr"].get(n): r[n.lower() + "_per_launch"] = avg(x["ctr"][n])
        if r.get("sq_lds_idx_active_per_launch"):
            r["lds_bank_conflict_frac_of_lds_cycles"] = r.get("sq_lds_bank_conflict_per_launch", 0.0) / r["sq_lds_idx_active_per_launch"]
        if r.get("sq_wave_cycles_per_launch") and r.get("sq_wait_inst_lds_per_launch") is not None:
            r["wait_inst_lds_frac_of_wave_cycles"] = r["sq_wait_inst_lds_per_launch"] / r["sq_wave_cycles_per_launch"]
        if r.get("insts_mfma_per_launch") and r.get("sq_insts_valu_per_launch"):
            r["valu_insts_per_mfma"] = (r["sq_insts_valu_per_launch"] - r["insts_mfma_per_launch"]) / r["insts_mfma_per_launch"]
            r["lds_insts_per_mfma"] = r.get("sq_insts_lds_per_launch", 0.0) / r["insts_mfma_per_launch"]
    y = match(s3, key)
    if y and y["ctr"].get("SQ_WAVE_CYCLES"):
        wc = avg(y["ctr"]["SQ_WAVE_CYCLES"])
        r["active_inst_frac_of_wave_cycles"] = {n[len("SQ_ACTIVE_INST_"):].lower(): avg(y["ctr"][n]) / wc
                                                for n in y["ctr"] if n.startswith("SQ_ACTIVE_INST_")}
    f, w = match(fe, key), match(wr, key)
    if f and w and f["ctr"].get("FETCH_SIZE") and w["ctr"].get("WRITE_SIZE"):
        r["hbm_bytes_per_launch"] = (2.0 * avg(f["ctr"]["FETCH_SIZE"]) + avg(w["ctr"]["WRITE_SIZE"])) * 1024.0
        r["hbm_gbs"] = r["hbm_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
    fl = {"self-attention (K6)": 4.0 * S * S * d, "cross-attention (K9)": 4.0 * S * 512 * d}.get(key[1])
    if fl: r["algorithmic_tflops"] = fl / (r["avg_us"] * 1e-6) / 1e12
    rows.append(r)
out = {"tag": tag, "tokens": S, "command": cmd + " (2 denoise steps incl. warm-up)",
       "notes": ["effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, both from the SAME counter pass",
                 "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", "hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction)",
                 "valu_insts_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA (the MFMAs are VALU instructions to the counter)",
                 "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves"],
       "kernels": rows}
json.dump(out, open(f"gpurun_out/prof_r06_{tag}.json", "w"), indent=1)
with open(f"gpurun_out/prof_r06_{tag}.md", "w") as o:
    o.write(f"`{cmd}`\n\n| kernel group | calls | avg us | % GPU time | alg. TF/s | eff. clock GHz | MFMA busy @ eff. clock | wave cycles: wait / issue-stall / active | VALU (non-MFMA) per MFMA | LDS insts per MFMA | LDS bank-conflict share | HBM GB/s | HBM MB/launch |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows[:16]:
        g = lambda k, fmt: (fmt % r[k]) if r.get(k) is not None else "-"
        ws = r.get("wave_cycles_split")
        o.write(f"| {r['what']} `{r['kernel'][:48]}` | {r['calls']} | {r['avg_us']:.1f} | {r['pct_of_gpu_time']:.1f} | {g('algorithmic_tflops', '%.0f')} | {g('effective_clock_ghz', '%.2f')} | "
                f"{g('mfma_busy_frac_at_effective_clock', '%.3f')} | {('%.2f / %.2f / %.2f' % (ws['wait_any'], ws['wait_inst_any'], ws['active_inst_any'])) if ws else '-'} | "
                f"{g('valu_insts_per_mfma', '%.2f')} | {g('lds_insts_per_mfma', '%.2f')} | {g('lds_bank_conflict_frac_of_lds_cycles', '%.3f')} | {g('hbm_gbs', '%.0f')} | "
                f"{('%.0f' % (r['hbm_bytes_per_launch'] / 1e6)) if r.get('hbm_bytes_per_launch') else '-'} |\n")
print(open(f"gpurun_out/prof_r06_{tag}.md").read())
PY
cp $D/stats/s_kernel_stats.csv gpurun_out/prof_r06_${TAG}_kernel_stats.csv 2>/dev/null
grep -h '"metric"' $D/bench_stats.log > gpurun_out/prof_r06_${TAG}_bench_line.json 2>/dev/null
tail -5 $D/sq3.log > gpurun_out/prof_r06_${TAG}_sq3_pass.log 2>/dev/null
rm -rf $D
