#!/usr/bin/env python3
"""Per-sweep durations from a cold start (HIP events around every sweep): is the first stretch of a run slower, and for
how long?  Usage: step_series.py [nsweeps=96] [X=65536] [Y=65536]   (ISING_FUSED=0/1 selects the launch form)"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
X = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
Y = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
t0 = time.perf_counter()
with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
    s.init()
    s.synchronize()
    t1 = time.perf_counter()
    ms = [s.sweep_timed(1) for _ in range(n)]
    t2 = time.perf_counter()
print(f"create+init {1e3 * (t1 - t0):.1f} ms; {n} sweeps {1e3 * (t2 - t1):.1f} ms wall")
for i in range(0, n, 8):
    print(f"sweeps {i + 1:3d}-{i + 8:3d}: " + " ".join(f"{v:6.3f}" for v in ms[i:i + 8]))
tail = sorted(ms[n // 2:])
print(f"median of the second half {tail[len(tail) // 2]:.4f} ms = {X * Y / tail[len(tail) // 2] * 1e-6:.1f} flips/ns; first 5: {sum(ms[:5]) / 5:.4f}; sweeps 6-25: {sum(ms[5:25]) / 20:.4f}")
