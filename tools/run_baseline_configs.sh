#!/bin/bash
# Runs the single-GPU BASELINE.json configurations through the cuIsing-compatible CLI and records the transcripts.
# Usage (on the GPU box): tools/run_baseline_configs.sh > gpurun_out/configs.txt
CLI=ising_gpu_amd/cuIsing
echo "### config 2: 16384x16384, T=Tc (-a 1), seed 1234, 10^5 sweeps (-p 10000)"
$CLI -x 16384 -y 16384 -d 1 -n 100000 -a 1 -s 1234 -p 10000 | grep -E "magnetization|Kernel execution|temp:|spins:"
echo
echo "### config 3: 65536x65536, T=Tc, 256 sweeps"
$CLI -x 65536 -y 65536 -d 1 -n 256 -a 1 -s 1234 -p 64 --energy | grep -E "magnetization|energy|Kernel execution|temp:|spins:"
echo
echo "### config 4 (one GPU's share of the 131072^2 / 8-GPU run): -x 131072 -y 16384, T=Tc, 256 sweeps"
$CLI -x 131072 -y 16384 -d 1 -n 256 -a 1 -s 1234 | grep -E "magnetization|Kernel execution|temp:|spins:"
echo
echo "### config 4 as 8 slabs on ONE device (decomposition check of the 131072x131072 lattice, 2 sweeps only)"
$CLI -x 131072 -y 16384 -d 8 --devmap 0,0,0,0,0,0,0,0 -n 2 -a 1 -s 1234 | grep -E "magnetization|Kernel execution|total lattice size"
echo
echo "### config 4 as 8 slabs on ONE device, 128 sweeps: the C-ABI ring (second stream per slab) at full length"
$CLI -x 131072 -y 16384 -d 8 --devmap 0,0,0,0,0,0,0,0 -n 128 -a 1 -s 1234 | grep -E "magnetization|Kernel execution|total lattice size"
echo
echo "### config 5: 8192x8192, seed 1234, T = 1.50 .. 3.00 step 0.05 in ONE process (--tsweep): per point a fresh lattice,"
echo "### 1000 equilibration sweeps, 100 measurements 10 sweeps apart: <|m|>, <m^2>, chi, U4, <e>, Cv"
$CLI -x 8192 -y 8192 -d 1 -s 1234 --tsweep 1.5,3.0,0.05,1000,100,10 | grep -E "^T = |Temperature sweep"
