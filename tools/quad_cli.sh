#!/bin/bash
# cuIsing on lattices the quad path takes (round 5) against the same library without it (ISING_QUAD=0), with and without the reference's -p 16 print points
# and with --energy; the checksum is of the magnetisation lines of a 64-sweep run (same in every form).   (profiles/quad_cli_r05.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/ising_gpu_amd/cuIsing
for cfg in "2048 2048 65536" "2048 512 131072" "4096 4096 16384" "2048 16384 16384" "6144 6144 8192" "8192 1024 32768"; do
  set -- $cfg
  for t in 1 0; do
    for p in "" "-p 16" "-p 16 --energy"; do
      r=$(ISING_QUAD=$t $C -x $1 -y $2 -n $3 -s 1234 -a 1 $p 2>&1 | grep -i "flips/ns" | tail -1 | sed 's/.*ms, //; s/(BW[^)]*)//')
      m=$(ISING_QUAD=$t $C -x $1 -y $2 -n 64 -s 1234 -a 1 -p 16 2>&1 | grep -i "magn" | md5sum | cut -c1-8)
      echo "$2 x $1, $3 sweeps, ISING_QUAD=$t ${p:-no prints}: $r   [-p 16 lines of a 64-sweep run: $m]"
    done
  done
done
# the reference's default run: no -x / -y
for t in 1 0; do ISING_QUAD=$t $C -n 65536 -p 16 2>&1 | grep -iE "flips/ns|Final" | sed "s/^/default lattice, ISING_QUAD=$t: /"; done
