#!/usr/bin/env python3
"""flips/ns of ising_sweep on the ballot layout over lattice shapes, default policy against tail strips off / fused on."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
shapes = [(16384, 16384), (32768, 32768), (65536, 65536), (131072, 16384), (16384, 131072), (65536, 16384), (16384, 65536), (32768, 8192), (8192, 32768), (24576, 24576)]
for X, Y in shapes:
    sweeps = max(16, min(1024, (1 << 34) // (X * Y) * 4))
    row = []
    for name, env in (("default", {}), ("no tail", {"ISING_TAIL": "0", "ISING_FUSED": "0"}), ("fused", {"ISING_FUSED": "1"})):
        for k in ("ISING_TAIL", "ISING_FUSED"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT) as s:
            s.init(); s.sweep_timed(max(8, sweeps // 4))
            best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
            row.append(f"{name}: {best:7.1f}" + (" (fused)" if s.fused and name == "default" else ""))
    print(f"{Y:6d} rows x {X:6d} cols (H={s.strip_rows})  " + "   ".join(row), flush=True)
