#!/usr/bin/env bash
# Build libicvideo.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Usage: infinicube_amd/csrc/build.sh [extra hipcc flags]
#        ICV_EXPERIMENTS=1 infinicube_amd/csrc/build.sh   also compiles the measured-slower A/B kernels under experiments/
#        (attention families 1, 3, 4, 5, 6, 9, the software-pipelined bf16 attention attn7q and the 4-wave GEMM) into the library; the shipped build leaves them out.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${ICV_LIB_OUT:-$here/libicvideo.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -I"$here" -Wno-unused-result)
objs=()
pids=()
mkdir -p "$here/build"
srcs=(api elementwise gemm gemm256 gemm256p gemm_fp8 fp8 attention attn2 attn7 attn7p attn8 buffers voxels vae_ops dit_forward comm ipc conv)
tag=""
if [[ "${ICV_EXPERIMENTS:-0}" == "1" ]]; then
  srcs+=(experiments/attn1 experiments/attn3 experiments/attn4 experiments/attn5 experiments/attn6 experiments/attn9 experiments/attn7q experiments/gemm256w experiments/gemm256x)
  FLAGS+=(-DICV_EXPERIMENTS)
  tag="x"          # separate object files: the two configurations differ in -DICV_EXPERIMENTS
fi
for src in "${srcs[@]}"; do
  obj="$here/build/$(basename "$src")$tag.o"
  if [[ ! -f "$obj" || "$here/$src.hip" -nt "$obj" || "$here/icv_common.h" -nt "$obj" || "$here/attn_common.h" -nt "$obj" || "$root/include/icvideo.h" -nt "$obj" ]]; then
    extra=()
    # buffers.hip produces BYTE outputs that must equal the reference's: no fused multiply-add contraction there
    [[ "$src" == buffers || "$src" == voxels ]] && extra=(-ffp-contract=off)
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" "$@" -c "$here/$src.hip" -o "$obj" &
    pids+=($!)
  fi
  objs+=("$obj")
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -ldl -o "$out"
python3 -c "import ctypes,sys; ctypes.CDLL(sys.argv[1])" "$out"   # unresolved symbols fail here, not on the GPU box
echo "built $out"
