mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05h_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05h_pytest.txt
timeout 600 python -m pytest tests/test_gpu_policy.py -q -s > gpurun_out/r05h_policy.txt 2>&1
timeout 300 python bench.py > gpurun_out/r05h_bench.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05h_bench_20_5.txt 2>&1
timeout 600 python tools/ab_probe.py --env ";ISING_SPLIT=0" --shapes 8192x8192,8192x16384,16384x16384,24576x24576,65536x8192,65536x16384,131072x2048,65536x1024,32768x4096 --H 0 --wgs 0 > gpurun_out/r05h_auto.txt 2>&1
