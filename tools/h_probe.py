#!/usr/bin/env python3
"""Single slabs whose strip-height rule stops short of 16 rows: fused launches at H = 4 / 8 / 16.  Usage: h_probe.py X Y [X Y ...]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

args = [int(v) for v in sys.argv[1:]]
with ig.IsingSlab(32768, 32768, seed=1, temp=ig.CRIT_TEMP_F32) as s:  # clock ramp
    s.init(); s.sweep_timed(256)
for X, Y in zip(args[::2], args[1::2]):
    sweeps = max(32, min(4096, (1 << 35) // (X * Y) * 8)) // 32 * 32
    out = []
    for H in (0, 4, 8, 16):
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, strip_rows=H) as s:
            s.init(); s.sweep_timed(32)
            out.append(f"H={s.strip_rows}{'*' if H == 0 else ''}: {max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3)):7.1f}")
    print(f"{Y:6d} x {X:6d}  " + "  ".join(out), flush=True)
