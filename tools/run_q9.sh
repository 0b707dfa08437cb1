mkdir -p gpurun_out/q9
P="python tools/quad_probe.py --shapes 4,8,12,64:4,8,12,64,2:4,8,12,64,1:4,8,12,128,2:4,8,12,32,2 2048 2048"
echo "== product, draws at low stream priority"; $P
echo "== product, ISING_QUAD_PRIO=0 (normal)"; ISING_QUAD_PRIO=0 $P
for w in 4 5 6; do echo "== draws at $w waves per SIMD at most, low priority"; ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qd$w.so $P; echo "== draws at $w waves per SIMD at most, normal priority"; ISING_QUAD_PRIO=0 ISING_LIB=$PWD/ising_gpu_amd/libising_hip_qd$w.so $P; done
