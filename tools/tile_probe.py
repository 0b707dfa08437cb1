#!/usr/bin/env python3
"""Small lattices on the dense layout: tile launches of several sweeps (ising_dense.hip: dense_tile_k, ISING_TILES=1 + ISING_TILE_*) against
one launch per colour -- full state compared after an uneven number of sweeps, then flips/ns over a timed run.
Usage: tile_probe.py [X Y ...]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

KEYS = ("ISING_TILES", "ISING_TILE_ROWS", "ISING_TILE_WORDS", "ISING_TILE_SWEEPS", "ISING_TILE_THREADS", "ISING_TILE_XCD")


def run(X, Y, env, check=None):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sweeps = max(512, (1 << 34) // (X * Y) // 64 * 64)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_DENSE) as s:
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
    return best, ok, state


SHAPES = None  # --shapes TR,TWI,S,NT:TR,TWI,S,NT...: these only
if len(sys.argv) > 2 and sys.argv[1] == "--shapes":
    SHAPES = [tuple(map(int, t.split(","))) for t in sys.argv[2].split(":")]
    del sys.argv[1:3]
sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(2048, 2048), (4096, 4096), (8192, 2048), (2048, 8192), (4096, 2048), (6144, 3072)]
for X, Y in sizes:
    base, _, ref = run(X, Y, {"ISING_TILES": "0"})
    print(f"{Y} x {X}: one launch per colour {base:7.1f} flips/ns = {X * Y / base * 1e-3:6.2f} us per sweep", flush=True)
    gx = X // 2048
    dflt, ok, _ = run(X, Y, {}, ref)
    print(f"  the library's choice: {dflt:7.1f} flips/ns  x {dflt / base:4.2f}  state {'==' if ok else '!='} per-colour launches", flush=True)
    rows = []
    for TWI, TR, S, NT in ([(b, a, c, d) for a, b, c, d in SHAPES] if SHAPES else itertools.product((8, 16, 32), (8, 16, 32, 64), (2, 3, 4, 6, 8), (256, 512, 1024))):
        if Y % TR or (gx * 32) % TWI:
            continue
        tiles = (gx * 32 // TWI) * (Y // TR)
        items = (TR + 2 * S) * (TWI + 2)
        if not SHAPES and (tiles < 128 or tiles > 2048 or items > 4 * NT or items * 3 < NT):
            continue
        env = {"ISING_TILES": "1", "ISING_TILE_ROWS": str(TR), "ISING_TILE_WORDS": str(TWI), "ISING_TILE_SWEEPS": str(S), "ISING_TILE_THREADS": str(NT)}
        try:
            f, ok, _ = run(X, Y, env, ref)
        except Exception as e:  # noqa: BLE001
            print(f"  {TR:2d} rows x {TWI} words, {S} sweeps, {NT:4d} threads: {e}", flush=True)
            continue
        rows.append((f, TR, TWI, S, NT, ok, tiles))
        print(f"  {TR:2d} rows x {TWI} words, {S} sweeps a launch, {NT:4d} threads, {tiles:4d} tiles: {f:7.1f} flips/ns  x {f / base:4.2f}  state {'==' if ok else '!='} per-colour launches", flush=True)
    rows.sort(reverse=True)
    if rows:
        print(f"  best: {rows[0]}", flush=True)
        f, TR, TWI, S, NT, ok, tiles = rows[0]
        env = {"ISING_TILES": "1", "ISING_TILE_ROWS": str(TR), "ISING_TILE_WORDS": str(TWI), "ISING_TILE_SWEEPS": str(S), "ISING_TILE_THREADS": str(NT), "ISING_TILE_XCD": "0"}
        f2, ok2, _ = run(X, Y, env, ref)
        print(f"  the same without the tiles dealt to the XCDs in bands (ISING_TILE_XCD=0): {f2:7.1f} flips/ns, state {'==' if ok2 else '!='}", flush=True)
