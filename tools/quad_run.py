#!/usr/bin/env python3
"""One lattice on the quad path for a profiler: quad_run.py X Y sweeps [calls]  (shape through ISING_QUAD_*; prints flips/ns of the last call)"""
import os
import sys
import time

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

X, Y, n = map(int, sys.argv[1:4])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.environ.setdefault("ISING_QUAD", "1")
with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_DENSE) as s:
    s.init().sweep(64)
    s.synchronize()
    for _ in range(calls):
        t0 = time.perf_counter()
        s.sweep(n)
        s.synchronize()
        dt = time.perf_counter() - t0
    print(f"{Y} x {X}: quad={s.quad} {X * Y * n / dt * 1e-9:.1f} flips/ns, {dt / n * 1e6:.2f} us per sweep, up/down {s.count()}")
