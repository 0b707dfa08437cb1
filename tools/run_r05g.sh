mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_ring_capi.py tests/test_gpu_fused.py -x -q > gpurun_out/r05g_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05g_pytest.txt
E="ISING_SPLIT=1 ISING_SPLIT_DEPTH=1;ISING_SPLIT=1 ISING_SPLIT_DEPTH=4"
timeout 1200 python tools/ab_probe.py --libs r05a,product --env "$E" --shapes 8192x8192 --H 4,8 --wgs 5,6 > gpurun_out/r05g_split.txt 2>&1
timeout 1200 python tools/ab_probe.py --libs r05a,product --env "$E" --shapes 8192x16384,16384x16384,65536x8192 --H 8,16 --wgs 5,6 >> gpurun_out/r05g_split.txt 2>&1
timeout 300 python tools/ab_probe.py --libs r05a,product --shapes 16384x16384,65536x8192,65536x65536 --H 0 --wgs 0 >> gpurun_out/r05g_split.txt 2>&1
timeout 200 bash tools/first_contact.sh 2 > gpurun_out/r05g_first_contact.txt 2>&1
timeout 300 python bench.py --gpus 2 --workload strong --steps 32 --warmup 32 > gpurun_out/r05g_bench2.txt 2>&1
