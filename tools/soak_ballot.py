#!/usr/bin/env python3
"""Soak test of the ballot kernel's scalar-store / write-back / L1-bypass protocol: a long run on the ballot layout next
to the same run on the dense layout (v_cmpx kernel, no scratch traffic); counts and bond sums must agree at every
checkpoint and the final states word for word.  Usage: soak_ballot.py [X Y sweeps checkpoints [seed]]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

X, Y, sweeps, cps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (65536, 65536, 10000, 20)))
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 777
slabs = {name: ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, layout=lay).init()
         for name, lay in (("ballot", ig.LAYOUT_BALLOT), ("dense", ig.LAYOUT_DENSE))}
t0 = time.time()
for k in range(cps):
    obs = {}
    for name, s in slabs.items():
        s.sweep(sweeps // cps)
        obs[name] = (s.count(), s.bond_equal())
    ok = obs["ballot"] == obs["dense"]
    print(f"after {slabs['dense'].it:6d} sweeps: {obs['dense']} {'==' if ok else '!='} ballot  [{time.time() - t0:.0f} s]", flush=True)
    if not ok:
        raise SystemExit(f"MISMATCH: {obs}")
for color in (ig.BLACK, ig.WHITE):
    a, b = slabs["ballot"].read(color), slabs["dense"].read(color)
    assert np.array_equal(a, b), f"colour {color}: final states differ in {(a != b).sum()} words"
print(f"soak ok: {X}x{Y}, {sweeps} sweeps, final states identical")
