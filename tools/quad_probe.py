#!/usr/bin/env python3
"""Small lattices on the quad path (ising_quad.hip: draws ahead of the lattice, word passes on tiles; ISING_QUAD=1 + ISING_QUAD_*) against the library's
choice without it -- full state compared after an uneven number of sweeps, then flips/ns over a timed run.
Usage: quad_probe.py [--shapes C,T,NW:...] [X Y ...]"""
import os
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

KEYS = ("ISING_QUAD", "ISING_QUAD_C", "ISING_QUAD_T", "ISING_QUAD_WAVES")


def run(X, Y, env, check=None, layout=ig.LAYOUT_DENSE):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sweeps = max(512, (1 << 34) // (X * Y) // 64 * 64)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=layout) as s:
        s.init().sweep(37)
        s.sweep(2)
        s.sweep(1)
        state = (s.read(ig.BLACK), s.read(ig.WHITE), s.count(), s.bond_equal())
        ok = check is None or (np.array_equal(state[0], check[0]) and np.array_equal(state[1], check[1]) and state[2:] == check[2:])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            s.sweep(64)
            s.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            s.sweep(sweeps)
            s.synchronize()
            best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
        what = "quad" if s.quad else ("tiles" if s.tiled else ("fused" if s.fused else "per colour"))
    return best, ok, state, what


SHAPES = [(4, 8, 12), (4, 8, 8), (4, 8, 16), (8, 8, 16), (2, 8, 12), (4, 4, 8), (4, 4, 12), (4, 4, 16), (4, 12, 16), (4, 6, 12), (8, 4, 12)]
if len(sys.argv) > 2 and sys.argv[1] == "--shapes":
    SHAPES = [tuple(map(int, t.split(","))) for t in sys.argv[2].split(":")]
    del sys.argv[1:3]
sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(2048, 2048), (4096, 4096), (2048, 8192), (8192, 2048), (4096, 2048)]
for X, Y in sizes:
    base, _, ref, what = run(X, Y, {"ISING_QUAD": "0"}, layout=ig.LAYOUT_AUTO)
    print(f"{Y} x {X}: the library without the quad path ({what}) {base:7.1f} flips/ns = {X * Y / base * 1e-3:6.2f} us per sweep", flush=True)
    rows = []
    for sh in SHAPES:
        C, T, NW = sh[:3]
        env = {"ISING_QUAD": "1", "ISING_QUAD_C": str(C), "ISING_QUAD_T": str(T), "ISING_QUAD_WAVES": str(NW)}
        try:
            f, ok, _, what = run(X, Y, env, ref)
        except Exception as e:  # noqa: BLE001
            print(f"  {sh}: {e}", flush=True)
            continue
        rows.append((f, sh, ok))
        print(f"  tiles of {C:2d} row groups, {T:2d} sweeps a pass, {NW:2d} waves {sh[3:]}: {f:7.1f} flips/ns  x {f / base:4.2f}  ({what}) state {'==' if ok else '!='} the library's", flush=True)
    rows.sort(reverse=True)
    if rows:
        print(f"  best: {rows[0]}", flush=True)
