#!/usr/bin/env python3
"""Cost of the slab-ring schedule (edge rows launch + row exchange + interior launch per colour) on ONE GPU: n slabs
of one device driven by LocalRing against the same lattice as a single slab.  Usage: ring_overhead_probe.py [X Ytot n sweeps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

X, Y, n, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (65536, 65536, 2, 64)))


def timed(fn):
    fn(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(sweeps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
    s.init()
    dt = timed(lambda k: (s.sweep(k), s.synchronize()))
    print(f"single slab {Y}x{X}: {X * Y * sweeps / dt * 1e-9:8.1f} flips/ns  layout {s.layout}")
backs = [ig.HipSlabBackend.create(X, Y // n, seed=1, temp=ig.CRIT_TEMP_F32, nslabs=n, slab=k) for k in range(n)]
ring = ig.LocalRing(backs).init()
dt = timed(lambda k: ring.sweep(k))
print(f"{n} slabs of {Y // n}x{X} on one device (edges + copies + interior): {X * Y * sweeps / dt * 1e-9:8.1f} flips/ns")
for b in backs:
    b.slab.close()
