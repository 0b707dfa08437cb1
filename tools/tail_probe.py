#!/usr/bin/env python3
"""Tail strips of plain ballot launches: flips/ns by (tail rows, tail strip height).  ISING_TAIL is read at slab creation."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
os.environ["ISING_FUSED"] = "0"
for n, H in ((8192, 4), (8192, 8), (16384, 8), (16384, 4), (32768, 8), (65536, 8)):
    sweeps = max(32, min(4096, (1 << 34) // (n * n) * 8))
    row = []
    for tail in ("", f"{n//32},1", f"{n//16},1", f"{n//8},1", f"{n//4},1", f"{n//8},2", f"{n//4},2", f"{n//4},4" if H > 4 else f"{n//2},2"):
        if tail:
            os.environ["ISING_TAIL"] = tail
        else:
            os.environ.pop("ISING_TAIL", None)
        with ig.IsingSlab(n, n, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
            s.init(); s.sweep_timed(max(8, sweeps // 8))
            best = max(n * n * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
            row.append(f"{tail or 'none'}:{best:7.1f}")
    print(f"{n}^2 H={H}  " + "  ".join(row), flush=True)
