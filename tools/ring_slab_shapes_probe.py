#!/usr/bin/env python3
"""Ring slabs of wide-and-short shapes (what a strong-scaling split of a mid-size lattice gives every rank): a ring of ONE such slab (its own edge rows travel through the
transport into its 64 ghost rows, fused launches between the exchanges) against the same slab on its own -- rate, and the state after 100 sweeps word for word.
profiles/ring_slab_shapes_r04.txt      Usage: ring_slab_shapes_probe.py [X Y ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402


def rate(obj, X, Y, sync):
    sweeps = max(256, (1 << 36) // (X * Y) // 32 * 32)
    obj.sweep(64)
    sync()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        obj.sweep(sweeps)
        sync()
        best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
    return best


sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(16384, 2048), (32768, 1024), (65536, 512), (24576, 1536), (131072, 1024), (65536, 1024), (8192, 4096), (16384, 8192)]
for X, Y in sizes:
    with ig.IsingSlab(X, Y, seed=77, temp=ig.CRIT_TEMP_F32) as lone:
        lone.init().sweep(100)
        ref = (lone.read(ig.BLACK), lone.read(ig.WHITE), lone.count())
        r0 = rate(lone, X, Y, lone.synchronize)
        h0 = lone.strip_rows
    slab = ig.IsingSlab(X, Y, seed=77, temp=ig.CRIT_TEMP_F32, ring_halo=True)
    ring = ig.SlabSet([slab])
    try:
        ring.init().sweep(100)
        same = np.array_equal(slab.read(ig.BLACK), ref[0]) and np.array_equal(slab.read(ig.WHITE), ref[1]) and ring.count() == ref[2]
        r1 = rate(ring, X, Y, slab.synchronize)
        print(f"{Y} x {X}: lone slab {r0:6.0f} flips/ns (H = {h0}); ring of one {r1:6.0f} (H = {slab.strip_rows}, {slab.ghost_ptrs(ig.BLACK)[0]} ghost rows, "
              f"{slab.max_sweeps_per_launch} sweeps between exchanges); state after 100 sweeps {'==' if same else '!='} the lone slab's", flush=True)
        assert same
    finally:
        ring.close()
