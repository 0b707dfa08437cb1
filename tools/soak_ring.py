#!/usr/bin/env python3
"""Long run of a ring of one (ghost rows, RCCL to itself; then peer copies on the comm stream) next to the same lattice as a
single slab: counts and bond sums at every checkpoint, final states word for word.  usage: soak_ring.py [X Y sweeps checkpoints]"""
import os, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: F401
import ising_gpu_amd as ig
X, Y, sweeps, cps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (65536, 65536, 4000, 8)))
for name in ("rccl", "copy"):
    if name == "copy":
        os.environ.update(ISING_RING_INLINE="0", ISING_RING_STORE="0")
    single = ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32).init()
    slab = ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32, ring_halo=True)
    ring = (ig.NativeRing(slab) if name == "rccl" else ig.SlabSet([slab])).init()
    t0 = time.time()
    for k in range(cps):
        single.sweep(sweeps // cps)
        ring.sweep(sweeps // cps)
        a, b = (single.count(), single.bond_equal()), (ring.count(), ring.bond_equal())
        print(f"{name}: after {single.it:6d} sweeps: {a} {'==' if a == b else '!='} ring  [{time.time() - t0:.0f} s]", flush=True)
        if a != b:
            raise SystemExit("MISMATCH")
    for color in (ig.BLACK, ig.WHITE):
        assert np.array_equal(single.read(color), slab.read(color))
    print(f"{name}: soak ok: {X}x{Y}, {sweeps} sweeps, final states identical", flush=True)
    if name == "rccl":
        ring.close()
    slab.close(); single.close()
