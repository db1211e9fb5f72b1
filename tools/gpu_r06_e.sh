#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=/root/repo:/root/repo/tests
mkdir -p gpurun_out
run() {  # label, env...
  local label=$1; shift
  port=$((29600 + RANDOM % 300))
  for r in 0 1; do
    env "$@" RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port OMP_NUM_THREADS=2 timeout 200 python tests/rank_worker.py --out /tmp/o_$label.pt \
      --backend gloo --share-gpu --ops hip --model small --frames 17 --height 128 --width 160 --scenario loop --parallelism sp --kv-exchange ${KV:-ipc} > /tmp/rank_${label}_$r.log 2>&1 &
  done
  wait
  echo "== $label: $(grep -h 'Error\|error' /tmp/rank_${label}_*.log | grep -v amdgpu | head -3 | tr '\n' ' ')"
}
run default A=1
run legacy_wait ICV_IPC_LEGACY_WAIT=1
run no_publish ICV_IPC_NO_PUBLISH=1
run both ICV_IPC_LEGACY_WAIT=1 ICV_IPC_NO_PUBLISH=1
run q16 GPU_MAX_HW_QUEUES=16
KV=allgather run allgather A=1
python - <<'PY'
import torch
a = torch.load("/tmp/o_allgather.pt")["result"]
for l in ("default", "legacy_wait", "no_publish", "both", "q16"):
    try:
        b = torch.load(f"/tmp/o_{l}.pt")["result"]
        print(l, "equal to allgather:", torch.equal(a, b), "finite", bool(torch.isfinite(b).all()))
    except Exception as e:
        print(l, "no result:", type(e).__name__)
PY
