mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tiles.py -q -x > gpurun_out/r05i_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05i_pytest.txt
for L in product r05a; do
  echo "== library $L"
  if [ $L = r05a ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_r05a.so; else unset ISING_LIB; fi
  timeout 900 python tools/tile_probe.py 2048 2048 4096 4096 2048 1024 4096 2048 1024 1024 2>&1
done > gpurun_out/r05i_tiles.txt
