#!/usr/bin/env bash
# round 5, lease B: the shifted-row convolution kernel + the padded-volume VAE: parity tests, then wall-clock vs MIOpen
set -uo pipefail
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vae_hip_gpu.py -q -s 2>&1 | grep -v "MIOpen(HIP)" | tail -60 | tee gpurun_out/r05b_vae_hip_tests.txt
ICV_VAE_CONV=hip timeout 900 python tools/aux_bench.py 2>&1 | grep -v "MIOpen(HIP)" | tee gpurun_out/r05b_vae_hip_bench.txt
