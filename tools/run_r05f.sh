mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_ring_capi.py tests/test_gpu_fused.py -x -q > gpurun_out/r05f_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05f_pytest.txt
E=";ISING_SPLIT=1 ISING_SPLIT_DEPTH=1;ISING_SPLIT=1 ISING_SPLIT_DEPTH=2;ISING_SPLIT=1 ISING_SPLIT_DEPTH=4"
timeout 1200 python tools/ab_probe.py --env "$E" --shapes 8192x8192 --H 4,8 --wgs 5,6 > gpurun_out/r05f_split.txt 2>&1
timeout 1200 python tools/ab_probe.py --env "$E" --shapes 8192x16384,16384x16384,65536x8192,24576x24576 --H 8,16 --wgs 5,6 >> gpurun_out/r05f_split.txt 2>&1
for cfg in "16384 16384 16 5 1" "16384 16384 16 5 4" "8192 8192 4 6 1" "8192 8192 4 6 4" "65536 8192 16 5 4"; do
  set -- $cfg
  echo "== $1 x $2 H=$3 wgs=$4/CU lead=1 depth=$5 (trace build)"
  ISING_LIB=$PWD/ising_gpu_amd/libising_hip_trace.so ISING_SPLIT=1 ISING_SPLIT_DEPTH=$5 ISING_FUSED_WGS=$((256*$4)) python tools/ab_probe.py case $1 $2 $3 2>&1
done > gpurun_out/r05f_trace.txt 2>&1
