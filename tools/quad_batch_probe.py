#!/usr/bin/env python3
"""Batches of small lattices on the quad path (ising_batch.cpp: batch_sweep_quad): n lattices of one shape in one quad_pass_k launch per pass, against one of
them alone -- every member compared with a lone run after an uneven number of sweeps, then aggregate flips/ns over a timed run, plain and with print points.
Usage: quad_batch_probe.py [--shapes C,T,NW:...] [X Y n ...]"""
import os
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

KEYS = ("ISING_QUAD", "ISING_QUAD_C", "ISING_QUAD_T", "ISING_QUAD_WAVES")


def timed(fn, flips):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        best = max(best, flips / (time.perf_counter() - t0) * 1e-9)
    return best


def lone(X, Y):
    for k in KEYS:
        os.environ.pop(k, None)
    sweeps = max(512, (1 << 33) // (X * Y) // 64 * 64)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init().sweep(40)
        ref = (s.read(ig.BLACK), s.read(ig.WHITE))
        what = "quad" if s.quad else ("tiles" if s.tiled else ("fused" if s.fused else "per colour"))

        def go():
            s.sweep(sweeps)
            s.synchronize()
        return timed(go, X * Y * sweeps), ref, what


def batch(X, Y, n, env, ref):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sweeps = max(128, (1 << 35) // (X * Y * n) // 64 * 64)
    temps = [ig.CRIT_TEMP_F32] + [1.5 + 1.5 * r / max(1, n - 1) for r in range(1, n)]
    slabs = [ig.IsingSlab(X, Y, seed=1234, temp=t) for t in temps]
    with ig.IsingBatch(slabs) as b:
        b.init().sweep(37).sweep(2).sweep(1)
        ok = ref is None or (np.array_equal(slabs[0].read(ig.BLACK), ref[0]) and np.array_equal(slabs[0].read(ig.WHITE), ref[1]))

        def go():
            b.sweep(sweeps)
            slabs[0].synchronize()
        plain = timed(go, X * Y * n * sweeps)
        counted = timed(lambda: b.sweep_counted(sweeps, 16, True), X * Y * n * sweeps)
        shape = b.quad_shape
    for s in slabs:
        s.close()
    return plain, counted, ok, shape


SHAPES = []
if len(sys.argv) > 2 and sys.argv[1] == "--shapes":
    SHAPES = [tuple(map(int, t.split(","))) for t in sys.argv[2].split(":")]
    del sys.argv[1:3]
cases = [tuple(map(int, sys.argv[i:i + 3])) for i in range(1, len(sys.argv), 3)] or [(2048, 2048, 31), (4096, 4096, 31), (2048, 2048, 8), (2048, 512, 31), (6144, 2048, 16), (8192, 1024, 31)]
for X, Y, n in cases:
    base, ref, what = lone(X, Y)
    print(f"{Y} x {X}: one lattice alone ({what}) {base:7.1f} flips/ns", flush=True)
    for sh in [None] + SHAPES:
        env = {} if sh is None else {"ISING_QUAD_C": str(sh[0]), "ISING_QUAD_T": str(sh[1]), "ISING_QUAD_WAVES": str(sh[2])}
        try:
            plain, counted, ok, shape = batch(X, Y, n, env, ref)
        except Exception as e:  # noqa: BLE001
            print(f"  {n} x, {sh}: {e}", flush=True)
            continue
        print(f"  {n:3d} lattices, shape (C, T, waves) {shape}{'' if sh is None else ' forced'}: {plain:7.1f} flips/ns, with counts and energy every 16: {counted:7.1f}  member 0 {'==' if ok else '!='} the lone run", flush=True)
