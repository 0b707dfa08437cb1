mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tiles.py -q -x > gpurun_out/r05k_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05k_pytest.txt
for L in product r05a; do
  echo "== library $L"
  if [ $L = r05a ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_r05a.so; else unset ISING_LIB; fi
  timeout 900 python tools/tile_probe.py --shapes 32,8,4,512:32,8,4,1024:32,8,6,1024:32,8,8,1024:64,16,4,1024:64,16,6,1024:32,16,4,1024:16,8,3,512:16,8,4,512:16,8,6,1024 2048 2048 4096 4096 2048 1024 4096 2048 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05k_tiles.txt
