#!/usr/bin/env python3
"""Small lattices: dense plain launches vs ballot plain vs ballot fused with 8-wave workgroups (flips/ns, best of 3)."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
for X, Y in ((8192, 2048), (8192, 4096), (8192, 8192), (16384, 4096), (16384, 8192), (8192, 16384), (16384, 16384)):
    sweeps = max(64, min(8192, (1 << 34) // (X * Y) * 8))
    row = []
    for name, lay, env, H in (("dense", ig.LAYOUT_DENSE, {}, 0), ("ballot plain", ig.LAYOUT_BALLOT, {"ISING_FUSED": "0"}, 0),
                              ("fused wide H=1", ig.LAYOUT_BALLOT, {"ISING_FUSED": "1", "ISING_FUSED_WIDE": "1"}, 1),
                              ("fused wide H=2", ig.LAYOUT_BALLOT, {"ISING_FUSED": "1", "ISING_FUSED_WIDE": "1"}, 2)):
        for k in ("ISING_FUSED", "ISING_FUSED_WIDE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=lay, strip_rows=H) as s:
            s.init(); s.sweep_timed(max(8, sweeps // 8))
            best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
            row.append(f"{name} (H={s.strip_rows}): {best:7.1f}")
    print(f"{Y:6d} x {X:6d}  " + "   ".join(row), flush=True)
