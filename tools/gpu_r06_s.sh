#!/usr/bin/env bash
# round 6 (s): gemm_fp8 with the conflict-free fragment chunk assignment: tests, A/B against the previous build on the same box, counters
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "fp8" 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | tee gpurun_out/r06_gemm_fp8_tests.txt
{ echo "== before (chunks 2kg, 2kg+1)"; ICV_LIB_PATH=$PWD/infinicube_amd/csrc/build/libicvideo_before.so timeout 300 python tools/gemm_fp8_bench.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids"; echo "== after (chunks kg, 4+kg)"; timeout 300 python tools/gemm_fp8_bench.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids"; } | tee gpurun_out/r06_gemm_fp8_ab.txt
FUZZ_FP8=1 timeout 300 python tools/fuzz_kernels.py 60 8 2>&1 | tail -2 | tee -a gpurun_out/r06_gemm_fp8_tests.txt
D=gpurun_out/prof_r06_gf8; rm -rf $D; mkdir -p $D
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D -o q -- python tools/gemm_fp8_bench.py > $D/run.log 2>&1
python - $D <<'PY' | tee gpurun_out/r06_gemm_fp8_counters.txt
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_fp8" not in n: continue
        key = (n[:44], r.get("Grid_Size") or r.get("Grid_Size_X"))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, c in sorted(rows.items()):
    a = {k: sum(v) / len(v) for k, v in c.items()}
    dur = a["dur_us"]; clk = a["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3
    print(f"{key[0]} grid {key[1]:>9s}: {dur:8.1f} us  eff clock {clk:.2f} GHz  MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (clk * 1e3 * dur):.3f}  LDS bank-conflict share {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
PY
rm -rf $D
