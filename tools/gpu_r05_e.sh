#!/usr/bin/env bash
# round 5, lease E: attention round trace; fp8 wire A/B at the N = 8 shard shape of config #5; the whole -m gpu suite
set -uo pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/attn_round_trace.py 2>&1 | tee gpurun_out/r05e_attn_round_trace.txt
cd /tmp && export TMPDIR=/tmp
for WIRE in bf16 e4m3; do
  ICV_FP8_WIRE=$WIRE MODEL=14b-i2v GRID=720p GEMM=fp8 ATTN=fp8 ONLY=8:4:pair ITERS=2 LAYERS=2 timeout 900 \
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wire_$WIRE -o w -- python $R/tools/sp_shard_compute_time.py > $R/gpurun_out/r05e_wire_$WIRE.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob
out = open("gpurun_out/r05e_fp8_wire_ab.txt", "w")
for wire in ("bf16", "e4m3"):
    f = glob.glob(f"gpurun_out/prof_wire_{wire}/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    pre = [r for r in rows if any(t in r["Name"] for t in ("amax_kernel", "quant_rows_kernel", "quant_vt_kernel"))]
    att = [r for r in rows if "attn8_kernel" in r["Name"]]
    line = [l for l in open(f"gpurun_out/r05e_wire_{wire}.log") if "ms per layer" in l]
    out.write(f"wire {wire}: {line[-1].strip() if line else '?'}\n")
    out.write(f"   quantise pre-pass (amax + quant_rows + quant_vt): {sum(float(r['TotalDurationNs']) for r in pre)/1e6:.2f} ms = {100*sum(float(r['TotalDurationNs']) for r in pre)/tot:.2f} % of the GPU time of the run; "
              f"e4m3 attention kernels {sum(float(r['TotalDurationNs']) for r in att)/1e6:.2f} ms = {100*sum(float(r['TotalDurationNs']) for r in att)/tot:.2f} %\n")
    for r in sorted(pre, key=lambda r: -float(r["TotalDurationNs"])):
        out.write(f"      {r['Name'][:70]:70s} calls {r['Calls']:>5} total {float(r['TotalDurationNs'])/1e6:8.2f} ms\n")
out.close()
print(open("gpurun_out/r05e_fp8_wire_ab.txt").read())
PY
rm -rf gpurun_out/prof_wire_bf16 gpurun_out/prof_wire_e4m3
timeout 2400 python -m pytest tests -m gpu -q --durations=25 2>&1 | grep -v "MIOpen(HIP)" | tail -60 | tee gpurun_out/r05e_gpu_suite.txt
