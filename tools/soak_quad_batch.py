#!/usr/bin/env python3
"""Soak of the batched quad passes (ising_batch.cpp: batch_sweep_quad) against the same lattices swept one by one (the lone quad path, itself held against the
oracle and the tile launches by tests/test_gpu_quad.py and tools/soak_quad.py): n lattices of one shape at n temperatures and two seeds, the same calls -- uneven
lengths, every third with print points (every 16; every other one of those with the energy) -- through the batch and through every member alone; final
states word for word, every print point of every member.
Usage: soak_quad_batch.py [X Y n sweeps ...]   (profiles/soak_quad_batch_r06.txt)"""
import hashlib
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402


def calls(sweeps):
    rng = np.random.default_rng(11)
    left, k, out = sweeps, 0, []
    while left:
        n = min(left, int(rng.integers(1, 2048)))
        out.append((n, k % 3 == 2, k % 2 == 0))
        left -= n
        k += 1
    return out


def setup(X, Y, n):
    temps = [1.5 + 1.5 * r / max(1, n - 1) for r in range(n)]
    seeds = [20260930 + (r % 2) for r in range(n)]
    return temps, seeds


def digest(s):
    return hashlib.sha256(s.read(ig.BLACK).tobytes() + s.read(ig.WHITE).tobytes()).hexdigest()[:16]


def run_batch(X, Y, n, plan):
    temps, seeds = setup(X, Y, n)
    slabs = [ig.IsingSlab(X, Y, seed=s, temp=t) for t, s in zip(temps, seeds)]
    series = [[] for _ in range(n)]
    with ig.IsingBatch(slabs) as b:
        assert b.quad_shape is not None
        b.init()
        t0 = time.perf_counter()
        for k, counted, energy in plan:
            if counted:
                for point in b.sweep_counted(k, 16, energy):
                    for r in range(n):
                        series[r].append(point[r])
            else:
                b.sweep(k)
        slabs[0].synchronize()
        dt = time.perf_counter() - t0
        out = [(digest(s), s.count(), s.bond_equal(), hashlib.sha256(repr(series[r]).encode()).hexdigest()[:12]) for r, s in enumerate(slabs)]
        shape = b.quad_shape
    for s in slabs:
        s.close()
    return out, dt, shape, sum(len(x) for x in series)


def run_alone(X, Y, n, plan):
    temps, seeds = setup(X, Y, n)
    out = []
    t0 = time.perf_counter()
    for t, sd in zip(temps, seeds):
        series = []
        with ig.IsingSlab(X, Y, seed=sd, temp=t) as s:
            assert s.quad
            s.init()
            for k, counted, energy in plan:
                if counted:
                    series += [p if energy else p[:2] + (None,) for p in s.sweep_counted(k, 16, energy)]
                else:
                    s.sweep(k)
            out.append((digest(s), s.count(), s.bond_equal(), hashlib.sha256(repr(series).encode()).hexdigest()[:12]))
    return out, time.perf_counter() - t0


cases = [tuple(map(int, sys.argv[i:i + 4])) for i in range(1, len(sys.argv), 4)] or [(2048, 2048, 31, 200000), (4096, 4096, 31, 40000), (2048, 512, 31, 400000), (6144, 2048, 16, 60000), (8192, 1024, 8, 100000)]
for X, Y, n, sweeps in cases:
    plan = calls(sweeps)
    a, dt_a, shape, npts = run_batch(X, Y, n, plan)
    b, dt_b = run_alone(X, Y, n, plan)
    same = a == b
    print(f"{n} x {Y} x {X}, {sweeps} sweeps in {len(plan)} calls of 1 .. 2047 ({npts} print points of all members among them), batch shape {shape}: batched {dt_a:6.1f} s "
          f"({X * Y * n * sweeps / dt_a * 1e-9:7.1f} flips/ns all in), one by one {dt_b:6.1f} s; every member's state, counts, bond sum and print points "
          f"{'==' if same else '!='} (member 0: {a[0][0]} / {b[0][0]})", flush=True)
    assert same, [(r, x, y) for r, (x, y) in enumerate(zip(a, b)) if x != y][:3]
