for lib in "" qkq "" qkq; do
  if [ -n "$lib" ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_$lib.so; else unset ISING_LIB; fi
  echo "==== ${lib:-product}"
  for s in "2048 2048 4096" "4096 4096 1024" "2048 16384 1024" "6144 6144 512"; do python tools/quad_run.py $s 3 2>&1 | grep flips; done
done
