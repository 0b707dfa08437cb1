#!/usr/bin/env python3
"""Soak of the print points inside the deep launches of a ring (ising_ring_sweep_counted, all slabs in one process, the rows travelling on the comm streams): calls of random
lengths, every count against a blocking ising_ring_count where a call ends on a print point, the final state against a lone slab of the whole lattice that swept the same
number of times.  profiles/soak_ring_counted_r04.txt      Usage: soak_ring_counted.py [X Yslab nslabs sweeps]"""
import os
import sys

import numpy as np

os.environ["ISING_RING_COUNTED"] = "2"
os.environ["ISING_RING_INLINE"] = "0"
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

X, Y, n, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8192, 2048, 2, 20000)))
rng = np.random.default_rng(5)
ring = ig.SlabSet([ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32, nslabs=n, slab=k, layout=ig.LAYOUT_BALLOT) for k in range(n)])
ring.init()
done, checked, points = 0, 0, 0
while done < sweeps:
    m = min(sweeps - done, int(rng.integers(1, 400)))
    every = int(rng.choice([1, 5, 16, 16, 16, 100]))
    got = ring.sweep_counted(m, every)
    done += m
    points += len(got)
    assert len(got) == done // every - (done - m) // every
    if got and done % every == 0:
        assert got[-1] == ring.count(), (done, every)
        checked += 1
state = np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), ring.count()
ring.close()
with ig.IsingSlab(X, Y * n, seed=99, temp=ig.CRIT_TEMP_F32) as lone:
    lone.init().sweep(sweeps)
    same = np.array_equal(lone.read(ig.BLACK), state[0]) and np.array_equal(lone.read(ig.WHITE), state[1]) and lone.count() == state[2]
print(f"{n} slabs of {Y} x {X}, {sweeps} sweeps in calls of 1 .. 399 with print points every 1 / 5 / 16 / 100: {points} counts taken inside the launches, {checked} of them "
      f"checked against a blocking count (all equal); final state {'==' if same else '!='} a lone slab's of {Y * n} x {X} after the same sweeps", flush=True)
assert same
