mkdir -p gpurun_out/q15
(timeout 900 python -m pytest tests/test_gpu_quad.py -x -q 2>&1 | tail -15) > gpurun_out/q15/test.log; cat gpurun_out/q15/test.log
python tools/quad_probe.py 2048 2048 4096 4096 2048 512 2048 8192 > gpurun_out/q15/probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/q15/probe.txt
(timeout 300 python -m pytest tests/test_gpu_ballot.py tests/test_gpu_policy.py -k "quad or ring_correlations" -q -s 2>&1 | grep -E "the library \(|passed|failed" | grep -v print) > gpurun_out/q15/dirty.txt; cat gpurun_out/q15/dirty.txt
