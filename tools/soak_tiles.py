#!/usr/bin/env python3
"""Soak of the tile launches (ising_dense.hip: dense_tile_k) against one launch per colour: the same lattice, seed and number of sweeps through both forms,
final states compared word for word (plus counts and bond sum); calls of uneven lengths so that launches of every length 1 .. S occur.
Usage: soak_tiles.py [X Y sweeps ...]   (profiles/soak_tiles_r04.txt)"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402


def run(X, Y, sweeps, tiles):
    os.environ["ISING_TILES"] = "1" if tiles else "0"
    rng = np.random.default_rng(7)
    with ig.IsingSlab(X, Y, seed=20260929, temp=ig.CRIT_TEMP_F32) as s:
        assert s.tiled == tiles
        s.init()
        t0 = time.perf_counter()
        left = sweeps
        while left:
            n = min(left, int(rng.integers(1, 4096)))
            s.sweep(n)
            left -= n
        s.synchronize()
        dt = time.perf_counter() - t0
        h = hashlib.sha256(s.read(ig.BLACK).tobytes() + s.read(ig.WHITE).tobytes()).hexdigest()[:16]
        return h, s.count(), s.bond_equal(), dt


cases = [tuple(map(int, sys.argv[i:i + 3])) for i in range(1, len(sys.argv), 3)] or [(2048, 2048, 1000000), (2048, 512, 1000000), (4096, 2048, 400000), (4096, 4096, 200000), (6144, 2048, 200000), (8192, 2048, 200000)]
for X, Y, sweeps in cases:
    a = run(X, Y, sweeps, True)
    b = run(X, Y, sweeps, False)
    print(f"{Y} x {X}, {sweeps} sweeps in calls of 1 .. 4095: tile launches {a[3]:6.1f} s ({X * Y * sweeps / a[3] * 1e-9:7.1f} flips/ns), one launch per colour {b[3]:6.1f} s; "
          f"state sha256 {a[0]} {'==' if a[:3] == b[:3] else '!='} {b[0]}, counts {a[1]}, bond sum {a[2]}", flush=True)
    assert a[:3] == b[:3]
