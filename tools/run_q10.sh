mkdir -p gpurun_out/q10
SH="4,8,12,32,2:4,8,8,32,2:4,8,12,64:8,8,16,32,2:2,8,12,32,2:4,4,8,32,2:2,6,8,36,2:4,4,16,32,2:8,4,16,32,2:4,12,16,48,2"
SIZES="2048 512 2048 1024 2048 4096 2048 8192 4096 1024 4096 2048 4096 4096 4096 8192 6144 2048 6144 6144 8192 1024 8192 2048 8192 4096 8192 8192 2048 16384 4096 16384"
for lib in qd4 ""; do
  if [ -n "$lib" ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_$lib.so; else unset ISING_LIB; fi
  echo "==== library: ${lib:-product}"
  python tools/quad_probe.py --shapes $SH $SIZES 2>&1 | grep -v amdgpu.ids
done
