#!/usr/bin/env python3
"""Cost of the slab-ring schedule (edge rows launch + halo delivery on the comm streams + interior launch per colour) on
ONE GPU: n slabs of one device through the C-ABI ring (ising_ring_sweep: on one device every launch writes its edge rows
straight into the neighbouring slabs' halo rows; ISING_RING_STORE=0: edge-row launch + copies + interior launch) and
through the torch-side LocalRing (edge launch, copies, interior launch on the compute stream), against the same lattice
as a single slab driven by per-colour launches and by fused launches.  Usage: ring_overhead_probe.py [X Ytot n sweeps]"""
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

X, Y, n, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (65536, 65536, 2, 64)))


def timed(fn, sync):
    fn(16)
    sync()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        fn(sweeps)
        sync()
        best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
    return best


for fused in ("0", "1"):
    os.environ["ISING_FUSED"] = fused
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        v = timed(lambda k: s.sweep(k), s.synchronize)
        print(f"single slab {Y}x{X}, {'fused' if s.fused else 'per-colour'} launches: {v:8.1f} flips/ns  layout {s.layout}")
        base = v if fused == "0" else base
ring = ig.SlabSet([ig.IsingSlab(X, Y // n, seed=1, temp=ig.CRIT_TEMP_F32, nslabs=n, slab=k) for k in range(n)]).init()
v = timed(lambda k: ring.sweep(k), ring.synchronize)
print(f"{n} slabs of {Y // n}x{X} on one device, C-ABI ring: {v:8.1f} flips/ns = {100 * (v / base - 1):+.2f} % vs per-colour single slab")
ring.close()
backs = [ig.HipSlabBackend.create(X, Y // n, seed=1, temp=ig.CRIT_TEMP_F32, nslabs=n, slab=k) for k in range(n)]
lring = ig.LocalRing(backs).init()
v = timed(lambda k: lring.sweep(k), torch.cuda.synchronize)
print(f"{n} slabs of {Y // n}x{X} on one device, torch-side LocalRing (copies on the compute stream):   {v:8.1f} flips/ns = {100 * (v / base - 1):+.2f} %")
for b in backs:
    b.slab.close()
