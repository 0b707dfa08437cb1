#!/usr/bin/env python3
"""Soak of the overlapped deep exchange: a ring of one slab over RCCL and over the IPC peer transport -- thousands of exchanges
next to running fused launches, edge strips polling the delivery counter -- against the same lattice as a lone slab in fused
launches; full state, counts and bond sums at every checkpoint.  A small slab on purpose: 32768 x 4096 is 1.3 ms per launch of
32 sweeps, so the exchange (0.45 ms of window on a big slab) has little time and the edge strips DO wait for it.
soak_overlap.py [X Y sweeps_total checkpoints]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

X, Y, total, ncheck = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32768, 4096, 64000, 8)))
step = total // ncheck
ref = ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT).init()
rings = {}
for tr in ("rccl", "ipc"):
    s = ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
    rings[tr] = (s, ig.NativeRing(s, transport=tr).init())
t0 = time.time()
for k in range(ncheck):
    ref.sweep(step)
    want = (ref.count(), ref.bond_equal())
    for tr, (s, ring) in rings.items():
        n = 0
        while n < step:  # uneven call lengths: launches of 32 and shorter ones, exchanges in between and at the ends
            m = min(step - n, (97, 32, 5, 64, 1, 33)[(n + k) % 6])
            ring.sweep(m)
            n += m
        got = (ring.count(), ring.bond_equal())
        ring.quiesce()
        same = np.array_equal(s.read(ig.BLACK), ref.read(ig.BLACK)) and np.array_equal(s.read(ig.WHITE), ref.read(ig.WHITE))
        print(f"{(k + 1) * step:7d} sweeps, {tr}: counts and bond sum {'==' if got == want else '!='} lone slab, state {'==' if same else '!='} [{time.time() - t0:.0f} s]", flush=True)
        assert got == want and same
for tr, (s, ring) in rings.items():
    ring.close()
    s.close()
ref.close()
print("soak ok")
