mkdir -p gpurun_out/q16
(timeout 900 python -m pytest tests/test_gpu_quad.py -x -q 2>&1 | tail -5) > gpurun_out/q16/test.log; cat gpurun_out/q16/test.log
python tools/quad_probe.py --shapes 4,8,12:4,8,8:4,8,16:8,8,16:2,8,12:4,12,16:4,6,12:4,8,12,2 2048 512 2048 2048 2048 8192 > gpurun_out/q16/probe1.txt 2>&1; grep -v amdgpu.ids gpurun_out/q16/probe1.txt
python tools/quad_probe.py --shapes 4,8,12:4,8,16:8,8,16:4,4,12:4,4,8:4,6,12:8,4,12:4,8,12,2 4096 1024 4096 4096 4096 16384 6144 2048 6144 6144 > gpurun_out/q16/probe2.txt 2>&1; grep -v amdgpu.ids gpurun_out/q16/probe2.txt
python tools/quad_probe.py --shapes 4,4,16:4,4,12:2,4,12:4,2,16:4,2,12:2,6,16 8192 1024 8192 2048 > gpurun_out/q16/probe3.txt 2>&1; grep -v amdgpu.ids gpurun_out/q16/probe3.txt
