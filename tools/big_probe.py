#!/usr/bin/env python3
"""Large lattices: plain launches (H, tail) against fused launches (H): flips/ns, best of 3."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
shapes = [(65536, 65536), (131072, 16384), (32768, 32768)]
for X, Y in shapes:
    sweeps = max(16, (1 << 34) // (X * Y) * 8)
    row = []
    for name, env, H in (("plain H=8 auto tail", {"ISING_FUSED": "0"}, 8), ("plain H=16 auto tail", {"ISING_FUSED": "0"}, 16),
                         ("plain H=16 tail 2048,2", {"ISING_FUSED": "0", "ISING_TAIL": "2048,2"}, 16), ("plain H=16 tail 4096,2", {"ISING_FUSED": "0", "ISING_TAIL": "4096,2"}, 16),
                         ("fused H=8", {"ISING_FUSED": "1"}, 8), ("fused H=16", {"ISING_FUSED": "1"}, 16), ("fused H=4", {"ISING_FUSED": "1"}, 4)):
        for k in ("ISING_FUSED", "ISING_TAIL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
            s.init(); s.sweep_timed(max(8, sweeps // 2))
            best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
            row.append(f"{name}: {best:7.1f}")
    print(f"{Y:6d} x {X:6d}  " + "   ".join(row), flush=True)
