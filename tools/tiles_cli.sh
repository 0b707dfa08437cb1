#!/bin/bash
# cuIsing on lattices that stay on the dense layout: tile launches (the default up to 2^24 spins) against one launch per colour (ISING_TILES=0),
# with and without the reference's -p 16 print points; the checksum is of the last magnetisation line of a 64-sweep run (same in every form).
# Run on the GPU box through gpurun: bash tools/tiles_cli.sh > gpurun_out/tiles_cli.txt   (profiles/tiles_cli_r04.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/ising_gpu_amd/cuIsing
for cfg in "2048 2048 16384" "2048 1024 16384" "4096 2048 8192" "4096 4096 4096" "2048 4096 8192" "6144 2048 8192" "8192 2048 4096"; do
  set -- $cfg
  for t in 1 0; do
    for p in "" "-p 16"; do
      r=$(ISING_TILES=$t $C -x $1 -y $2 -n $3 -s 1234 $p 2>&1 | grep -i "flips/ns" | tail -1 | sed 's/.*ms, //; s/(BW[^)]*)//')
      m=$(ISING_TILES=$t $C -x $1 -y $2 -n 64 -s 1234 -p 16 2>&1 | grep -i "magn" | tail -1 | md5sum | cut -c1-8)
      echo "$2 x $1, $3 sweeps, ISING_TILES=$t ${p:-no prints}: $r   [last -p line of a 64-sweep run: $m]"
    done
  done
done
