#!/usr/bin/env bash
# L2 hit rate of gemm256 (and of the vendor library's kernel on the same operands) on ONE DiT shape: one rocprofv3 counter
# pass (--kernel-trace + --pmc only).  Backs the byte budget of DESIGN.md §9 (round 3) with a measurement.
export TMPDIR=/tmp
out=gpurun_out/gemm_l2; rm -rf $out; mkdir -p $out
SHAPE=${1:-"37440,13824,5120,1"}
cat > /tmp/gemm_l2_one.py <<PY
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
for _ in range(6):
    ops.gemm(a, w, bias, out, epi, **kw)
torch.cuda.synchronize()
for _ in range(6):
    torch.nn.functional.linear(a, w)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/l2 -o q -- python /tmp/gemm_l2_one.py > $out/l2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/ea -o q -- python /tmp/gemm_l2_one.py > $out/ea.log 2>&1
python - "$SHAPE" <<'PY'
import csv, glob, collections, re, sys
M, N, K, epi = [int(x) for x in sys.argv[1].split(",")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("l2", "ea"):
    for f in glob.glob(f"gpurun_out/gemm_l2/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/gemm_l2_summary.txt", "w") as o:
    o.write(f"shape M={M} N={N} K={K} epilogue {epi}: L2 (TCC) counters per launch, rocprofv3 --pmc in its own passes\n")
    for k, c in acc.items():
        if not any(t in k for t in ("gemm256", "Cijk")):
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        hit, miss = m.get("TCC_HIT_sum"), m.get("TCC_MISS_sum")
        line = f"{k}: " + " | ".join(f"{n} {v:.4g}" for n, v in m.items())
        if hit is not None and miss is not None and hit + miss > 0:
            line += f" | L2 hit rate {hit / (hit + miss):.3f}"
        o.write(line + "\n"); print(line)
    tiles = ((M + 255) // 256) * (N // 256)
    o.write(f"operand bytes through the L2 -> LDS path: {tiles} tiles x {K // 64} K-tiles x 64 KiB = {tiles * (K // 64) * 65536 / 1e9:.1f} GB per launch; "
            f"a 4 x 8 patch of tiles per XCD shares 12 operand panels among 64 panel reads: at best {1 - 12 / 64:.3f} of those bytes hit\n")
PY
tail -2 $out/l2.log
