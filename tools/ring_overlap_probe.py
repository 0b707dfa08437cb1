#!/usr/bin/env python3
"""What ONE GPU of a ring pays for the deep exchange, with and without the round-3 changes: a ring of one slab (its own
first / last 64 rows travel through the transport into its ghost rows) against the same slab sweeping itself in fused
launches.  One variant per process (the switches are read once):
  ring_overlap_probe.py single|rccl|ipc X Y sweeps     with ISING_RING_OVERLAP=0 / ISING_RING_TRAPEZOID=0 in the environment
                                                       for the schedules of round 2
  ring_overlap_probe.py all X Y sweeps                 runs the table (subprocesses)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
X, Y, sweeps = (int(v) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (65536, 65536, 128)))

if mode == "all":
    rows = [("single slab, fused launches", "single", {}),
            ("ring of one, RCCL, overlapped + trapezoid (default)", "rccl", {}),
            ("ring of one, IPC, free-running overlap (no stream wait)", "ipc", {"ISING_RING_OVERLAP": "2"}),
            ("ring of one, RCCL, overlapped, all ghost rows every level", "rccl", {"ISING_RING_TRAPEZOID": "0"}),
            ("ring of one, RCCL, exchange between launches + trapezoid", "rccl", {"ISING_RING_OVERLAP": "0"}),
            ("ring of one, RCCL, round 2's schedule", "rccl", {"ISING_RING_OVERLAP": "0", "ISING_RING_TRAPEZOID": "0"}),
            ("ring of one, IPC peer transport, overlapped + trapezoid", "ipc", {}),
            ("ring of one, IPC peer transport, round 2's schedule", "ipc", {"ISING_RING_OVERLAP": "0", "ISING_RING_TRAPEZOID": "0"}),
            ("single slab, fused launches (again: drift of the box over the table)", "single", {})]
    base = None
    for name, m, env in rows:
        r = subprocess.run([sys.executable, __file__, m, str(X), str(Y), str(sweeps)], env=dict(os.environ, **env), capture_output=True, text=True)
        try:
            v = float([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1].split()[1])
        except (ValueError, IndexError):
            print(f"{name}: FAILED {r.stderr[-400:]}")
            continue
        base = base or v
        print(f"{Y}x{X} {name:62s} {v:8.1f} flips/ns  {100 * (v / base - 1):+.2f} %", flush=True)
    sys.exit(0)

import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402


def timed(fn, sync):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:  # clock warm-up
        fn(32)
        sync()
    best = 0.0
    for _ in range(4):
        t0 = time.perf_counter()
        fn(sweeps)
        sync()
        best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
    return best


if mode == "single":
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        print("RESULT", timed(lambda k: s.sweep(k), s.synchronize))
else:
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
        ring = ig.NativeRing(s, transport=mode).init()
        v = timed(lambda k: ring.sweep(k), ring.quiesce)
        ring.close()
        print("RESULT", v)
