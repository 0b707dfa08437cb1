#!/usr/bin/env python3
"""flips/ns of the update kernels by lattice size, device layout and strip height (rows per wave per launch).
Usage: strip_probe.py [sizes=8192,16384,32768,65536] [layouts=ballot,dense] [H=1,2,4,8,16]"""
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8192,16384,32768,65536").split(",")]
layouts = (sys.argv[2] if len(sys.argv) > 2 else "ballot,dense").split(",")
hs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,4,8,16").split(",")]
LAY = {"ballot": ig.LAYOUT_BALLOT, "dense": ig.LAYOUT_DENSE, "nibble": ig.LAYOUT_NIBBLE, "auto": ig.LAYOUT_AUTO}
for n in sizes:
    sweeps = max(32, min(4096, (1 << 34) // (n * n) * 8))
    for lay in layouts:
        row = []
        for h in hs:
            try:
                with ig.IsingSlab(n, n, seed=1234, temp=ig.CRIT_TEMP_F32, layout=LAY[lay], strip_rows=h) as s:
                    s.init()
                    s.sweep_timed(max(8, sweeps // 8))
                    best = 0.0
                    for _ in range(3):
                        ms = s.sweep_timed(sweeps)
                        best = max(best, n * n * sweeps / (ms * 1e6))
                    row.append(f"H={s.strip_rows}:{best:7.1f}")
            except ig.IsingError as e:
                row.append(f"H={h}:err")
        print(f"{n:6d}^2 {lay:7s} ({sweeps} sweeps)  " + "  ".join(row), flush=True)
