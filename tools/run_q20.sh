mkdir -p gpurun_out/q20
(timeout 900 python -m pytest tests/test_gpu_quad.py -x -q 2>&1 | tail -5) > gpurun_out/q20/test.log; cat gpurun_out/q20/test.log
python tools/quad_probe.py --shapes 4,8,12:4,8,8:4,8,16:2,8,12:4,12,16:4,4,12:4,6,12:8,4,12:8,8,16:4,4,16:2,6,16 2048 512 2048 1024 2048 2048 2048 4096 2048 8192 2048 16384 4096 1024 4096 2048 4096 4096 4096 16384 6144 2048 6144 6144 8192 1024 > gpurun_out/q20/probe.txt 2>&1; grep -v "amdgpu.ids\|too many items" gpurun_out/q20/probe.txt | grep -E "library without|best"
