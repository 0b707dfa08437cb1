#!/usr/bin/env python3
"""Many independent lattices in the same launches (ising_batch_*): a small temperature series from the ordered start, <|m|> and <e>
per temperature -- what `cuIsing --tsweep 1.8,2.1,0.1,400,20,10 --tsweep-cold` does in C++."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ising_gpu_amd as ig

X = Y = 8192
temps = [1.8, 1.9, 2.0, 2.1]
lattices = [ig.IsingSlab(X, Y, seed=100 + k, temp=t, layout=ig.LAYOUT_BALLOT) for k, t in enumerate(temps)]
all_up = np.full((Y, X // 64), 0xFFFFFFFF, dtype=np.uint32)             # the 1-bit host format: every spin up
for lat in lattices:
    lat.init()
    for colour in (ig.BLACK, ig.WHITE):
        lat.write_bits(colour, all_up)
with ig.IsingBatch(lattices) as batch:
    batch.sweep(400)                                                    # equilibration: one fused launch carries all four lattices
    for _ in range(20):
        batch.sweep(10).measure_enqueue()                               # measurements queue up behind the sweeps on the device
    series = batch.measure_fetch()                                      # [[(up, down, bond_equal) per lattice] per measurement]
n = float(X * Y)
for k, t in enumerate(temps):
    m = np.mean([abs(p[k][0] - p[k][1]) / n for p in series])
    e = np.mean([ig.energy_per_spin(p[k][2], X * Y) for p in series])
    yang = (1.0 - np.sinh(2.0 / t) ** -4) ** 0.125
    print(f"T = {t:.2f}: <|m|> = {m:.5f} (exact {yang:.5f}), <e> = {e:.5f}")
for lat in lattices:
    lat.close()
