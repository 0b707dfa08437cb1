#!/usr/bin/env python3
"""Soak of the quad path (ising_quad.hip) against the tile launches / one launch per colour of the same library (ISING_QUAD=0): the same lattice, seed and
number of sweeps through both, final states compared word for word (plus counts and bond sum); calls of uneven lengths so that passes of every length occur,
every third call with print points (ising_sweep_counted, every 16; every other one of those with the energy).
Usage: soak_quad.py [X Y sweeps ...]   (profiles/soak_quad_r05.txt)"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402


def run(X, Y, sweeps, quad):
    os.environ["ISING_QUAD"] = "1" if quad else "0"
    rng = np.random.default_rng(7)
    counts = []
    with ig.IsingSlab(X, Y, seed=20260930, temp=ig.CRIT_TEMP_F32) as s:
        assert s.quad == quad
        s.init()
        t0 = time.perf_counter()
        left, k = sweeps, 0
        while left:
            n = min(left, int(rng.integers(1, 4096)))
            if k % 3 == 2:
                counts += s.sweep_counted(n, 16, k % 2 == 0)  # (every other one with the bond sum at the print points)
            else:
                s.sweep(n)
            left -= n
            k += 1
        s.synchronize()
        dt = time.perf_counter() - t0
        h = hashlib.sha256(s.read(ig.BLACK).tobytes() + s.read(ig.WHITE).tobytes()).hexdigest()[:16]
        return h, s.count(), s.bond_equal(), hashlib.sha256(repr(counts).encode()).hexdigest()[:12], len(counts), dt


cases = [tuple(map(int, sys.argv[i:i + 3])) for i in range(1, len(sys.argv), 3)] or [(2048, 2048, 1000000), (2048, 512, 1000000), (4096, 4096, 200000), (2048, 16384, 100000), (6144, 6144, 60000), (8192, 1024, 200000)]
for X, Y, sweeps in cases:
    a = run(X, Y, sweeps, True)
    b = run(X, Y, sweeps, False)
    print(f"{Y} x {X}, {sweeps} sweeps in calls of 1 .. 4095 ({a[4]} print points among them): quad path {a[5]:6.1f} s ({X * Y * sweeps / a[5] * 1e-9:7.1f} flips/ns all in), "
          f"without {b[5]:6.1f} s; state sha256 {a[0]} {'==' if a[:5] == b[:5] else '!='} {b[0]}, counts {a[1]}, bond sum {a[2]}, print points {a[3]}", flush=True)
    assert a[:5] == b[:5]
