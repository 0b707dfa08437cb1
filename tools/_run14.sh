export ISING_ABORT_POLLS=40000
mkdir -p gpurun_out/r04i
ISING_FUSED_TICKET_ASYNC=1 timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_batch.py tests/test_gpu_ballot.py tests/test_gpu_random.py tests/test_gpu_ring_ipc.py -x -q -p no:cacheprovider 2>&1 | tail -4
timeout 1500 python tools/wait_late_probe.py 8192 4096 8192 8192 16384 8192 16384 16384 32768 32768 65536 65536 2>&1 | tee gpurun_out/r04i/ticket_async_probe.txt
