mkdir -p gpurun_out; export TMPDIR=/tmp
python -m infinicube_amd.videogen.test_api --synthetic --model 1.3b --frames 17 --height 256 --width 448 --steps 10 2>&1 | grep -v amdgpu | tail -12
