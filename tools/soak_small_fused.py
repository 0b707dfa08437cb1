#!/usr/bin/env python3
"""Soak of the fused launches of one-row units on lattices below 1.5 * 2^24 spins (the library's choice since the end of round 4) against the dense layout:
the same lattice, seed and sweeps through both, final states (the reference's packed words), counts and bond sums compared.  profiles/soak_small_fused_r04.txt
Usage: soak_small_fused.py [X Y sweeps ...]"""
import hashlib
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402


def run(X, Y, sweeps, layout):
    rng = np.random.default_rng(11)
    with ig.IsingSlab(X, Y, seed=20260929, temp=ig.CRIT_TEMP_F32, layout=layout) as s:
        how = f"{'ballot' if s.layout == ig.LAYOUT_BALLOT else 'dense'}{' fused H=%d' % s.strip_rows if s.fused else ''}{' tiles' if s.tiled else ''}"
        s.init()
        t0 = time.perf_counter()
        left = sweeps
        while left:
            n = min(left, int(rng.integers(1, 20000)))
            s.sweep(n)
            left -= n
        s.synchronize()
        dt = time.perf_counter() - t0
        h = hashlib.sha256(s.read(ig.BLACK).tobytes() + s.read(ig.WHITE).tobytes()).hexdigest()[:16]
        return (h, s.count(), s.bond_equal()), dt, how


cases = [tuple(map(int, sys.argv[i:i + 3])) for i in range(1, len(sys.argv), 3)] or [
    (8192, 1280, 400000), (8192, 2048, 400000), (8192, 2560, 300000), (16384, 768, 300000), (16384, 1152, 300000), (16384, 2048, 200000), (24576, 896, 200000),
    (32768, 640, 200000), (32768, 1024, 200000), (65536, 512, 100000), (12288, 1536, 200000), (6144, 3072, 200000)]
for X, Y, sweeps in cases:
    a, ta, ha = run(X, Y, sweeps, ig.LAYOUT_AUTO)
    b, tb, hb = run(X, Y, sweeps, ig.LAYOUT_DENSE)
    print(f"{Y} x {X}, {sweeps} sweeps in calls of 1 .. 19999: library ({ha}) {ta:5.1f} s = {X * Y * sweeps / ta * 1e-9:6.0f} flips/ns, dense layout ({hb}) {tb:5.1f} s; "
          f"state sha256 {a[0]} {'==' if a == b else '!='} {b[0]}, counts {a[1]}, bond sum {a[2]}", flush=True)
    assert a == b
