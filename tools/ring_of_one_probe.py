#!/usr/bin/env python3
"""What ONE GPU of a ring pays for the ring schedule: a ring of one slab (ring_halo: the slab's own edge rows travel
through the transport into its halo rows) against the same slab sweeping itself with per-colour launches.
Rows: copy transport with the copies on the compute stream (ISING_RING_INLINE=1: edge rows, copies, interior in one
stream), copy transport on the comm stream and RCCL send/recv to itself (both: edge rows + delivery on the comm stream
next to the interior rows on the compute stream).  Usage: ring_of_one_probe.py [X Y sweeps]"""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

X, Y, sweeps = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (65536, 65536, 64)))


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


os.environ["ISING_FUSED"] = "0"
with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
    s.init()
    base = timed(lambda k: s.sweep(k), s.synchronize)
print(f"single slab {Y}x{X}, per-colour launches: {base:8.1f} flips/ns")
for name, inline in (("copies on the compute stream", "1"), ("edge rows + copies on the comm stream", "0")):
    os.environ["ISING_RING_INLINE"] = inline
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
        ring = ig.SlabSet([s]).init()
        v = timed(lambda k: ring.sweep(k), ring.synchronize)
        print(f"ring of one, copy transport, {name}: {v:8.1f} flips/ns = {100 * (v / base - 1):+.2f} %")
with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
    ring = ig.NativeRing(s).init()
    v = timed(lambda k: ring.sweep(k), ring.quiesce)
    print(f"ring of one, RCCL send/recv to itself, edge rows + delivery on the comm stream: {v:8.1f} flips/ns = {100 * (v / base - 1):+.2f} %")
    ring.close()
