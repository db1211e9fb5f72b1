mkdir -p gpurun_out/r04d; export TMPDIR=/tmp
ITERS=4 python tools/sp_attn_shapes.py 2>&1 | grep -E "chunks  |matrix|one launch|---|full" | tee gpurun_out/r04d/sp_attn_shapes_kv.txt
for cfg in 4:4 8:4:pair; do
  tag=${cfg//:/_}
  (cd /tmp && ONLY=$cfg LAYERS=2 ITERS=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python $GRAFT_REPO_ROOT/tools/sp_shard_compute_time.py) > gpurun_out/r04d/trace_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python - "$f" gpurun_out/r04d/kernel_trace_$tag.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
with open(sys.argv[2], "w") as f:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        f.write(f"{v[1]/tot*100:5.1f}% n={v[0]:4d} avg={v[1]/v[0]:9.1f}us grid={k[1]} wg={k[2]} {k[0]}\n")
PY
done
head -14 gpurun_out/r04d/kernel_trace_4_4.txt
