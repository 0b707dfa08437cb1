#!/usr/bin/env python3
"""Batched fused launches against the same lattices one at a time and two side by side on private streams (round 2's
--tsweep): batch_probe.py [X Y n sweeps]  -> flips/ns (all lattices), plus the cost of one batched measurement."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

X, Y, n, sweeps = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8192, 8192, 31, 320)))
temps = [1.5 + 1.5 * r / max(1, n - 1) for r in range(n)]


def best(fn, sync, reps=3):
    fn()
    sync()
    v = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        v = max(v, X * Y * n * sweeps / (time.perf_counter() - t0) * 1e-9)
    return v


slabs = [ig.IsingSlab(X, Y, seed=1234, temp=t) for t in temps]
print(f"{n} lattices of {Y}x{X}, layout {slabs[0].layout}, lone strip rows {slabs[0].strip_rows}")
for s in slabs:
    s.init()


def lone():
    for s in slabs:
        s.sweep(sweeps)


print(f"one after the other (one stream):        {best(lone, slabs[0].synchronize):8.1f} flips/ns")
with ig.IsingBatch(slabs) as b:
    print(f"batch: strips of {b.strip_rows} rows, {b.wg_per_cu} workgroups per CU")
    v = best(lambda: b.sweep(sweeps), slabs[0].synchronize)
    print(f"batched fused launches:                  {v:8.1f} flips/ns")
    slabs[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        b.measure_enqueue()
    m = b.measure_fetch()
    dt = (time.perf_counter() - t0) / 100
    print(f"one batched measurement (count + bond sum of {n} lattices): {dt * 1e6:7.1f} us; first {m[0][0]}")
    t0 = time.perf_counter()
    for _ in range(20):
        for s in slabs:
            s.measure_enqueue()
    for s in slabs:
        s.measure_fetch()
    dt = (time.perf_counter() - t0) / 20
    print(f"the same through {n} x ising_measure_enqueue:               {dt * 1e6:7.1f} us")
for s in slabs:
    s.close()
