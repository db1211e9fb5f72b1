#!/usr/bin/env bash
# round 6 (p): the counters of att8::attn8_kernel again, now the software-pipelined loop (t2v 480p, e4m3 mode)
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 bash tools/gpu_prof_r06.sh t2v480_fp8_pipelined 37440 -- --model 14b --gemm-dtype fp8 --attn-dtype fp8 > gpurun_out/r06_prof_t2v480_fp8_pipelined.log 2>&1
tail -25 gpurun_out/r06_prof_t2v480_fp8_pipelined.log
D=gpurun_out/prof_r06_attn8_valu; rm -rf $D; mkdir -p $D
CMD="python bench.py --model 14b --gemm-dtype fp8 --attn-dtype fp8 --steps 1 --warmup 0 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_F8 SQ_BUSY_CU_CYCLES --output-format csv -d $D -o v -- $CMD > $D/run.log 2>&1
python - $D <<'PY' | tee gpurun_out/r06_attn8_valu_counters_pipelined.txt
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn8_kernel" in n or "gemm_fp8" in n:
            acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n, {k: sum(v) / len(v) for k, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
rm -rf $D
