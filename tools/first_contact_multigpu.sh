#!/usr/bin/env bash
# FIRST CONTACT WITH AN N-GPU NODE, in one go.  Every multi-GPU statement of this repo so far comes from gloo
# multi-process runs on CPU, N processes sharing one GPU and one-GPU rehearsals: RCCL / SDMA have never carried a K|V row over xGMI.
# This script produces, under profiles/<round>/ (default r06), everything a reader needs to judge the first real run:
#   first_contact_bench_n<N>_{auto,sp,auto_fp8}.json      python bench.py --gpus N: ONE JSON line each - the plan that ran (multi_gpu.plan,
#                                                .failed_attempts), the K|V autotune table PER CANDIDATE (transport x {arrival-driven,
#                                                4, 2 chunks}): two-layer time, raw exchange time -> receive GB/s and its fraction of the
#                                                7 x 153 GB/s links, self-attention UNDER the real exchange next to the same launches
#                                                served from memory, and for the copy-engine transport whether a pull needs compute
#                                                units (icv_ipc_probe_copy: "copy engine" vs "blit kernel")
#   first_contact_scaling.txt                    N = 1, 2, 4, 8 back to back -> steps/s and x of one GPU (the driver computes its own)
#   first_contact_copy_kernels.txt               rocprofv3 --kernel-trace over a short ipc run: per rank process, how many
#                                                __amd_rocclr_copyBuffer (blit) dispatches there were - the probe's independent check
#   first_contact_rccl_tests.txt                 pytest tests/test_multigpu_rccl.py -m gpu: the 19 tests that move bytes between two
#                                                devices and skip on a 1-GPU box
#   first_contact_generate_n<N>.json             ICV_WORLD=N behind the UNCHANGED caller (tools/e2e_wallclock.py): whole generate() calls,
#                                                the pool's plan record (every rank's GPU_MAX_HW_QUEUES / IPC mode), the ranks' autotune
# Usage: tools/first_contact_multigpu.sh [N=all visible GPUs] [round=r06]     (one node; ~15 min at N = 8)
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
R=${2:-r06}
OUT=profiles/$R
mkdir -p "$OUT" gpurun_out
SHARE=()
if [[ "${ICV_BENCH_SHARE_GPU:-0}" == "1" ]]; then SHARE=(ICV_DIST_BACKEND=gloo); fi     # 1-GPU rehearsal of this script: N ranks share the GPU over gloo
MODEL=${MODEL:-14b}
EXTRA=(${EXTRA_BENCH_ARGS:-})                 # e.g. "--frames 17 --height 128 --width 160" for a quick rehearsal of this script
TRACE_MODEL=${TRACE_MODEL:-1.3b}
echo "== first contact: N = $N ranks, model $MODEL, records under $OUT"
for layout in auto sp auto_fp8; do          # auto_fp8: config #5's e4m3 mode (e4m3 blobs on the wire; its "+arrival" candidates gate the chunk launches in the kernel)
  DT=(); [[ $layout == auto_fp8 ]] && DT=(--gemm-dtype fp8 --attn-dtype fp8)
  env "${SHARE[@]}" timeout 1500 python bench.py --gpus "$N" --model "$MODEL" "${EXTRA[@]}" "${DT[@]}" --steps 5 --warmup 2 --parallelism ${layout%_fp8} --no-cpu-baseline \
    2> "$OUT/first_contact_bench_n${N}_${layout}.stderr.txt" | tail -1 > "$OUT/first_contact_bench_n${N}_${layout}.json"
  python - "$OUT/first_contact_bench_n${N}_${layout}.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print("  no JSON line:", e); sys.exit(0)
mg = d.get("multi_gpu") or {}
print(f"  {d.get('value')} steps/s, {d.get('ms_per_step')} ms/step; plan {mg.get('plan')}; failed attempts {len(mg.get('failed_attempts') or [])}; error {d.get('error')}")
for r in ((mg.get("autotune") or {}).get("table") or []):
    print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if v is not None})
PY
done
if [[ "${ICV_BENCH_SHARE_GPU:-0}" != "1" ]]; then
  : > "$OUT/first_contact_scaling.txt"
  base=""
  for n in 1 2 4 8; do
    [[ $n -gt $N ]] && break
    line=$(timeout 1500 python bench.py --gpus $n --model "$MODEL" "${EXTRA[@]}" --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
    v=$(python -c "import json,sys; print(json.loads(sys.argv[1]).get('value') or 0)" "$line" 2>/dev/null || echo 0)
    [[ -z "$base" ]] && base=$v
    python -c "print(f'N = $n: {float($v):.4f} steps/s, x{float($v)/max(float($base),1e-12):.2f} of one GPU')" | tee -a "$OUT/first_contact_scaling.txt"
  done
fi
# independent check of the copy-engine probe: blit kernels in a kernel trace of a short ipc run
D=gpurun_out/first_contact_trace; rm -rf $D
env "${SHARE[@]}" timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python bench.py --gpus "$N" --model "$TRACE_MODEL" "${EXTRA[@]}" --steps 1 --warmup 1 \
  --parallelism sp --kv-exchange ipc --no-fallback --no-cpu-baseline > "$OUT/first_contact_copy_kernels.bench.txt" 2>&1
python - $D > "$OUT/first_contact_copy_kernels.txt" <<'PY'
import collections, csv, glob, sys
print("per traced process: dispatches of runtime copy (blit) kernels vs the transport's own flag kernels, in a 2-step sp/ipc run")
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    c = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("copyBuffer", "wait_ready_kernel", "wait_done_kernel", "publish_kernel", "streamOps", "attn7p_kernel", "attn7_kernel"):
            if key in n:
                c[key] += 1
    if c:
        print(f, dict(c))
print("copyBuffer > the number of own-rows / latent copies => the peers' rows moved by blit kernels, not by SDMA")
PY
rm -rf $D
[[ "${SKIP_RCCL_TESTS:-0}" == "1" ]] || timeout 3000 python -m pytest tests/test_multigpu_rccl.py -m gpu -q 2>&1 | grep -v "MIOpen(HIP)" | tail -30 > "$OUT/first_contact_rccl_tests.txt"
tail -3 "$OUT/first_contact_rccl_tests.txt" 2>/dev/null
env "${SHARE[@]}" ICV_WORLD=$N ${ICV_BENCH_SHARE_GPU:+ICV_TEST_SHARE_GPU=1} MODEL=$MODEL STEPS=${STEPS:-50} timeout 2400 python tools/e2e_wallclock.py 2> "$OUT/first_contact_generate_n${N}.stderr.txt" \
  | tail -1 > "$OUT/first_contact_generate_n${N}.json"
tail -c 600 "$OUT/first_contact_generate_n${N}.json"; echo
echo "== done: $(ls $OUT/first_contact_* | wc -l) files under $OUT"
