mkdir -p gpurun_out/q19
(timeout 900 python -m pytest tests/test_gpu_quad.py -x -q 2>&1 | tail -5) > gpurun_out/q19/test.log; cat gpurun_out/q19/test.log
python tools/quad_probe.py 2048 512 2048 2048 2048 8192 4096 1024 4096 4096 4096 16384 6144 2048 6144 6144 8192 1024 8192 2048 > gpurun_out/q19/probe.txt 2>&1; grep -v "amdgpu.ids\|too many items" gpurun_out/q19/probe.txt
