#!/usr/bin/env python3
"""A/B of the fused launch form's latency-hiding switches (variant builds, ISING_LIB): flips/ns, best of 3."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
os.environ["ISING_FUSED"] = "1"
for X, Y, wide, H in ((8192, 8192, 1, 1), (8192, 16384, 1, 1), (16384, 16384, 1, 1), (16384, 16384, 1, 2), (16384, 16384, 0, 4), (16384, 16384, 0, 8),
                      (32768, 32768, 0, 8), (65536, 65536, 0, 8)):
    os.environ["ISING_FUSED_WIDE"] = str(wide)
    sweeps = max(64, min(8192, (1 << 34) // (X * Y) * 8))
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        s.init(); s.sweep_timed(max(8, sweeps // 8))
        best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
    print(f"{Y:6d} x {X:6d} wide={wide} H={H}: {best:7.1f}", flush=True)
