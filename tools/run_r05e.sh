mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05e_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05e_pytest.txt
timeout 300 python bench.py > gpurun_out/r05e_bench.txt 2>&1
timeout 1500 python tools/ab_probe.py --env ";ISING_SPLIT=1 ISING_SPLIT_LEAD=1;ISING_SPLIT=1 ISING_SPLIT_LEAD=2" --shapes 8192x4096,8192x8192,8192x16384,16384x16384,24576x24576,32768x32768,65536x8192,65536x16384,65536x65536 --H 0,4,8,16 --wgs 0,5,6 > gpurun_out/r05e_split.txt 2>&1
