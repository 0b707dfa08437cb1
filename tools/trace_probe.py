#!/usr/bin/env python3
"""Where a fused launch's workgroup time goes (needs a -DISING_FUSED_TRACE variant build via ISING_LIB)."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
os.environ["ISING_FUSED"] = "1"
cases = ((8192, 8192, 1, 1), (16384, 16384, 1, 1), (16384, 16384, 1, 2), (16384, 16384, 0, 4), (65536, 65536, 0, 8))
for X, Y, wide, H in cases:
    os.environ["ISING_FUSED_WIDE"] = str(wide)
    sweeps = max(64, min(8192, (1 << 34) // (X * Y) * 8))
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        s.init()
        ms = s.sweep_timed(sweeps)
        print(f"{Y:6d} x {X:6d} wide={wide} H={H}: {X * Y * sweeps / (ms * 1e6):7.1f} flips/ns (trace build), {ms * 1e3 / (2 * sweeps):8.2f} us per colour", flush=True)
    sys.stdout.flush()
