mkdir -p gpurun_out
python tools/ab_probe.py --env "ISING_SPLIT=1 ISING_SPLIT_LEAD=1;ISING_SPLIT=1 ISING_SPLIT_LEAD=2;ISING_SPLIT=1 ISING_SPLIT_LEAD=3" --shapes 16384x16384 --H 8,16 --wgs 4,5,6 > gpurun_out/r05d_split.txt 2>&1
python tools/ab_probe.py --env ";ISING_SPLIT=1 ISING_SPLIT_LEAD=1;ISING_SPLIT=1 ISING_SPLIT_LEAD=2" --shapes 8192x8192,16384x8192 --H 2,4,8 --wgs 4,5,6 >> gpurun_out/r05d_split.txt 2>&1
for cfg in "16384 16384 16 5" "16384 16384 16 6" "16384 16384 8 6" "16384 16384 8 5" "8192 8192 4 5"; do
  set -- $cfg
  echo "== $1 x $2 H=$3 wgs=$4/CU lead=1 (trace build)"
  ISING_LIB=$PWD/ising_gpu_amd/libising_hip_trace.so ISING_SPLIT=1 ISING_FUSED_WGS=$((256*$4)) python tools/ab_probe.py case $1 $2 $3 2>&1
done > gpurun_out/r05d_trace.txt 2>&1
