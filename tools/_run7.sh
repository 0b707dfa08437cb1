export ISING_ABORT_POLLS=40000
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_batch.py tests/test_gpu_ballot.py tests/test_gpu_random.py -x -q -p no:cacheprovider 2>&1 | tail -6
mkdir -p gpurun_out/r04e
timeout 900 python tools/wait_late_probe.py 8192 8192 16384 16384 2>&1 | tee gpurun_out/r04e/wait_late_probe2.txt
