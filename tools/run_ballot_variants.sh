#!/bin/bash
# A/B builds of the ballot kernel (make -C ising_gpu_amd/csrc variant NAME=.. DEFS=..), timed on the bench lattice
for v in "" $@; do
  if [ -n "$v" ]; then export ISING_LIB=$PWD/ising_gpu_amd/libising_hip_$v.so; else unset ISING_LIB; fi
  echo "== variant '$v'"; python tools/perf_probe.py 65536 65536 64 8 0 3 2>&1 | tail -1
done
