#!/usr/bin/env python3
"""One lattice on one MI355X through the Python mirror of the C-ABI (include/ising_hip.h): what `cuIsing -x 16384 -y 16384 -a 1 -n 256`
does, plus the observables the reference does not have.  Run from the repo root after `make -C ising_gpu_amd/csrc`."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ising_gpu_amd as ig

X = Y = 16384
with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as lattice:   # T = T_c; the library picks device layout and launch form
    lattice.init()                                                      # latticeInit_k: spins from the seed, as the reference draws them
    up, down = lattice.count()
    print(f"start: m = {ig.magnetization(up, down):.6f}, e = {ig.energy_per_spin(lattice.bond_equal(), X * Y):.6f}")
    ms = lattice.sweep_timed(256)                                       # 256 lattice sweeps (black + white half-sweeps)
    up, down = lattice.count()
    print(f"after {lattice.it} sweeps: up {up}, down {down}, e = {ig.energy_per_spin(lattice.bond_equal(), X * Y):.6f}, "
          f"{X * Y * 256 / (ms * 1e6):.0f} flips/ns")
    corr = lattice.correlations(8)                                      # -c: exact integer sums, distance 1 .. 8
    print("C(r) =", " ".join(f"{c / (2.0 * X * Y):.4f}" for c in corr))
    with tempfile.TemporaryDirectory() as tmp:                          # binary checkpoint, 1 bit per spin, independent of the slab count
        ring = ig.SlabSet([lattice])
        ring.it = lattice.it
        ring.checkpoint_save(os.path.join(tmp, "lattice.ckpt"))
        print("checkpoint:", ig.checkpoint_info(os.path.join(tmp, "lattice.ckpt")))
