# round 3, job G: recorded runs — rocprofv3 stats + PMC passes at HEAD, e2e generate() with stage breakdown, config #2 at 50 steps
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_prof_r03.sh 14b > gpurun_out/prof_r03_14b.log 2>&1; tail -20 gpurun_out/prof_r03_14b.log
timeout 900 python tools/e2e_wallclock.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/e2e_generate_14b_r03.txt
ICV_SLOW_TESTS=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "config2_wan_1p3b_50_steps" 2>&1 | grep -E "config #2|passed|failed" | tail -5
