for mode in 1 2 3; do
  echo "==== ISING_QUAD_MODE=$mode"
  ISING_QUAD_MODE=$mode python tools/quad_probe.py --shapes 4,8,12:4,8,8:8,4,12 2048 2048 2048 8192 4096 4096 4096 16384 2>&1 | grep -v "amdgpu.ids\|too many items\|best"
done
ISING_QUAD_MODE=1 python tools/quad_probe.py --shapes 4,4,12 6144 6144 | grep -v amdgpu; ISING_QUAD_MODE=2 python tools/quad_probe.py --shapes 4,4,12 6144 6144 | grep "tiles of"; ISING_QUAD_MODE=3 python tools/quad_probe.py --shapes 4,4,12 6144 6144 | grep "tiles of"
