export TMPDIR=/tmp; out=gpurun_out/r04j3; rm -rf $out; mkdir -p $out
cat > /tmp/gemm_sw.py <<PY
import sys, os, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from infinicube_amd.videogen.ops import HipOps
M, N, K, epi = 37440, 13824, 5120, 1
ops = HipOps("cuda:0")
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn((N,), device="cuda")
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
ops.lib.icv_set_option(b"gemm256", 1)
ref = None
for sched, sw in ((3, 0), (67, 0), (67, 1), (67, 2), (67, 3)):
    ops.lib.icv_set_option(b"gemm256_sched", sched); ops.lib.icv_set_option(b"gemm256_ablate", sw << 2)
    ops.gemm(a, w, bias, out, epi); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    ok = torch.equal(out, ref)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(a, w, bias, out, epi)
    e1.record(); torch.cuda.synchronize()
    print(f"sched {sched} swizzle {sw}: {2.0 * M * N * K / (e0.elapsed_time(e1) / 5) / 1e9:.1f} TF/s  bit-identical {ok}", flush=True)
PY
python /tmp/gemm_sw.py 2>&1 | grep sched | tee $out/ksplit_swizzles.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/lds -o q -- python /tmp/gemm_sw.py > $out/lds.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r04j3/lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][20:52], float(r["Counter_Value"])))
rows.sort()
with open("gpurun_out/r04j3/ksplit_swizzles.txt", "a") as o:
    for i, (d, k, v) in enumerate(rows):
        if i % 6 == 0:
            line = f"dispatch group {i // 6} ({k}): SQ_LDS_BANK_CONFLICT {v:.4g}"
            print(line); o.write(line + "\n")
PY
