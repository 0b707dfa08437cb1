mkdir -p gpurun_out/r04d
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "sweep_graphs" -p no:cacheprovider 2>&1 | tail -15
timeout 900 python tools/small_probe.py 2>&1 | tee gpurun_out/r04d/small_probe.txt
