#!/usr/bin/env bash
# Round-5 profile of `bench.py` (one timed denoise step + one warm-up step) for profiles/r05/:
#   pass 1  rocprofv3 --kernel-trace --stats                     -> per-kernel time, self/cross attention and GEMM shapes split
#   pass 2  --pmc GRBM_GUI_ACTIVE + SQ issue/stall/MFMA counters  -> effective clock and MFMA-busy AT THAT CLOCK per kernel group
#   pass 3/4 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)      -> HBM bytes per launch (FETCH_SIZE x2: gfx950 correction)
# Counter passes use --kernel-trace + --pmc only (never with other trace domains).
# Usage: tools/gpu_prof_r05.sh [model=14b]   -> gpurun_out/prof_r05_<model>.json + .md
export TMPDIR=/tmp
M=${1:-14b}
D=gpurun_out/prof_r05_$M; rm -rf $D; mkdir -p $D
CMD="python bench.py --model $M --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o s -- $CMD > $D/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $D/sq -o q -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o w -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $D/sq2 -o q -- $CMD > /dev/null 2>&1
python - "$M" "$D" <<'PY'
import collections, csv, glob, json, re, sys
model, D = sys.argv[1], sys.argv[2]
S = 37440
DIMS = {"14b": (5120, 13824, 40), "1.3b": (1536, 8960, 12)}.get(model)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:64]
def keep(n): return any(t in n for t in ("attn", "gemm", "ln_modulate", "rmsnorm", "patchify", "unpatchify", "quantize"))
def label(name, grid, dur_us):
    """Human label of a dispatch group: self- vs cross-attention by duration, GEMM shapes by (epilogue, grid, duration)."""
    if "attn" in name:
        return "self-attention (K6)" if dur_us > 3000 else "cross-attention (K9)"
    if "gemm256p" in name and DIMS:      # persistent launch (one work-group per CU): the grid says nothing about N; epilogue + duration do
        d, f, _ = DIMS
        epi = re.search(r"gemm256p_kernel<(\d)", name)
        epi = int(epi.group(1)) if epi else -1
        if epi == 1: return f"FFN1 GEMM + GELU [2S,{d}]x[{f},{d}] (K10), persistent"
        if epi == 0:
            qkv = dur_us > 2.0 * (2 * S) * d * (2 * d) / 1.3e15 * 1e6       # midway between N = d and N = 3d at ~1.3 PF/s
            return f"QKV GEMM [2S,{d}]x[{3*d},{d}] (K4), persistent" if qkv else f"cross-q GEMM [2S,{d}]x[{d},{d}] (K9), persistent"
        return f"gemm256p epi {epi}"
    if "gemm256" in name and DIMS:
        d, f, _ = DIMS
        epi = re.search(r"gemm256_kernel<(\d)", name)
        epi = int(epi.group(1)) if epi else -1
        N, rows = None, "S"
        for mult, tag in ((2, "2S"), (1, "S")):        # 2S rows = the CFG-batched forward pair (the single-rank default)
            tiles_m = (mult * S + 255) // 256
            if grid % (512 * tiles_m) == 0 and (grid // 512 // tiles_m) * 256 in (d, 3 * d, f):
                N, rows = (grid // 512 // tiles_m) * 256, tag
                break
        if epi == 0 and N == 3 * d: return f"QKV GEMM [{rows},{d}]x[{3*d},{d}] (K4)"
        if epi == 0 and N == d: return f"cross-q GEMM [{rows},{d}]x[{d},{d}] (K9)"
        if epi == 1: return f"FFN1 GEMM + GELU [{rows},{d}]x[{f},{d}] (K10)"
        if epi == 2 and N == d: return f"O / cross-O / FFN2 GEMM + gated residual, {rows} rows, N={d} (K7/K9/K10)"
        return f"gemm256 epi {epi} N={N}"
    return name
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
fe = groups(f"{D}/fetch/**/*counter_collection.csv", True)
wr = groups(f"{D}/write/**/*counter_collection.csv", True)
s2 = groups(f"{D}/sq2/**/*counter_collection.csv", True)
tot = sum(sum(v["dur"]) for v in st.values())
rows = []
avg = lambda v: sum(v) / len(v) if v else None
for key, v in sorted(st.items(), key=lambda kv: -sum(kv[1]["dur"])):
    r = {"kernel": key[0], "what": key[1], "calls": len(v["dur"]), "avg_us": avg(v["dur"]), "pct_of_gpu_time": 100 * sum(v["dur"]) / tot}
    q = sq.get(key)
    if q and q["ctr"].get("GRBM_GUI_ACTIVE"):
        gui = avg(q["ctr"]["GRBM_GUI_ACTIVE"]) / 8.0          # the counter sums the 8 XCDs' GRBMs
        r["pmc_pass_avg_us"] = avg(q["dur"])
        r["effective_clock_ghz"] = gui / (avg(q["dur"]) * 1e3)
        mf = avg(q["ctr"].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
        r["mfma_busy_frac_at_effective_clock"] = mf / (1024.0 * gui) if gui else None
        r["mfma_busy_frac_at_2p4ghz"] = mf / (1024.0 * avg(q["dur"]) * 1e3 * 2.4)
        wc = avg(q["ctr"].get("SQ_WAVE_CYCLES", [0]))
        if wc:
            r["wave_cycles_split"] = {k: avg(q["ctr"].get(n, [0])) / wc for k, n in
                                      (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"))}
        r["insts_mfma_per_launch"] = avg(q["ctr"].get("SQ_INSTS_MFMA", [0]))
    f, w = fe.get(key), wr.get(key)
    if f and w and f["ctr"].get("FETCH_SIZE") and w["ctr"].get("WRITE_SIZE"):
        r["hbm_bytes_per_launch"] = (2.0 * avg(f["ctr"]["FETCH_SIZE"]) + avg(w["ctr"]["WRITE_SIZE"])) * 1024.0
        r["hbm_gbs"] = r["hbm_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
    x = s2.get(key)
    if x:
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if x["ctr"].get(n): r[n.lower() + "_per_launch"] = avg(x["ctr"][n])
    if DIMS:
        d, ffn, _ = DIMS
        fl = {"self-attention (K6)": 4.0 * S * S * d, "cross-attention (K9)": 4.0 * S * 512 * d}.get(key[1])
        rows_mult = 2.0 if "[2S," in key[1] else 1.0
        if "QKV GEMM" in key[1]: fl = 6.0 * S * d * d * rows_mult
        if "cross-q GEMM" in key[1]: fl = 2.0 * S * d * d * rows_mult
        if "FFN1" in key[1]: fl = 2.0 * S * d * ffn * rows_mult
        if fl: r["algorithmic_tflops"] = fl / (r["avg_us"] * 1e-6) / 1e12
    rows.append(r)
out = {"model": model, "command": f"python bench.py --model {model} --steps 1 --warmup 1 --no-cpu-baseline (2 denoise steps incl. warm-up)",
       "notes": ["effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, both from the SAME counter pass",
                 "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", "hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction)"],
       "kernels": rows}
json.dump(out, open(f"gpurun_out/prof_r05_{model}.json", "w"), indent=1)
with open(f"gpurun_out/prof_r05_{model}.md", "w") as o:
    o.write(f"| kernel group | calls | avg us | % GPU time | alg. TF/s | eff. clock GHz | MFMA busy @ eff. clock | MFMA busy @ 2.4 GHz | HBM GB/s | HBM MB/launch |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows[:14]:
        g = lambda k, fmt: (fmt % r[k]) if r.get(k) is not None else "-"
        o.write(f"| {r['what']} `{r['kernel'][:40]}` | {r['calls']} | {r['avg_us']:.1f} | {r['pct_of_gpu_time']:.1f} | {g('algorithmic_tflops', '%.0f')} | {g('effective_clock_ghz', '%.2f')} | "
                f"{g('mfma_busy_frac_at_effective_clock', '%.3f')} | {g('mfma_busy_frac_at_2p4ghz', '%.3f')} | {g('hbm_gbs', '%.0f')} | {('%.0f' % (r['hbm_bytes_per_launch'] / 1e6)) if r.get('hbm_bytes_per_launch') else '-'} |\n")
print(open(f"gpurun_out/prof_r05_{model}.md").read())
PY
cp $D/stats/s_kernel_stats.csv gpurun_out/prof_r05_${M}_kernel_stats.csv 2>/dev/null
grep -h '"metric"' $D/bench_stats.log > gpurun_out/prof_r05_${M}_bench_line.json 2>/dev/null
rm -rf $D
