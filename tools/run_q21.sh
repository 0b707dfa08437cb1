for lib in v706 ""; do
  if [ -n "$lib" ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_$lib.so; else unset ISING_LIB; fi
  echo "==== ${lib:-new}"
  python tools/quad_probe.py --shapes 4,8,12:4,8,8:4,4,12:8,4,12 2048 2048 2048 8192 4096 4096 4096 16384 6144 6144 2>&1 | grep -v "amdgpu.ids\|too many items"
done
