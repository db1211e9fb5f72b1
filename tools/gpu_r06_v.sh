#!/usr/bin/env bash
# round 6 (v): compute-only projection of the N-GPU step for config #5's form (Wan2.1-14B-I2V, 720p, e4m3 mode) and for the 480p e4m3 mode
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
MODEL=14b-i2v GRID=720p GEMM=fp8 ATTN=fp8 LAYERS=2 timeout 900 python tools/sp_shard_compute_time.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee gpurun_out/r06_sp_projection_config5_i2v720_fp8.txt
cp gpurun_out/sp_compute_only_projection.json gpurun_out/r06_sp_projection_config5_i2v720_fp8.json
true
true
