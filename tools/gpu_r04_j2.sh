# why is the k-split schedule slower?  LDS bank conflicts and L2 request counts of sched 3 vs 67 on the FFN1 shape (separate --pmc passes)
export TMPDIR=/tmp; out=gpurun_out/r04j2; rm -rf $out; mkdir -p $out
cat > /tmp/gemm_two.py <<PY
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
for sched in (3, 67):
    ops.lib.icv_set_option(b"gemm256_sched", sched)
    for _ in range(5):
        ops.gemm(a, w, bias, out, epi)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/lds -o q -- python /tmp/gemm_two.py > $out/lds.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -o q -- python /tmp/gemm_two.py > $out/l2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/tcp -o q -- python /tmp/gemm_two.py > $out/tcp.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04j2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r04j2/summary.txt", "w") as o:
    for k, c in acc.items():
        line = k + ": " + " | ".join(f"{n} {sum(v) / len(v):.4g}" for n, v in sorted(c.items()))
        print(line); o.write(line + "\n")
PY
