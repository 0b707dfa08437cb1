#!/usr/bin/env python3
"""Grid search over (strip height, tail rows, tail strip height) of plain ballot launches; prints the best few per shape."""
import os, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
os.environ["ISING_FUSED"] = "0"
shapes = [(int(a), int(b)) for a, b in (s.split("x") for s in (sys.argv[1:] or ["16384x16384", "32768x32768", "8192x8192"]))]
for X, Y in shapes:
    sweeps = max(16, min(2048, (1 << 34) // (X * Y) * 4))
    res = []
    for H in (2, 4, 8, 16):
        tails = sorted({0} | {Y // d // H * H for d in (64, 32, 16, 12, 8, 6, 4, 3)})
        for tail in tails:
            for h2 in ((1, 2) if tail and H > 2 else (1,)):
                if tail and (h2 >= H or 2 * tail >= Y):
                    continue
                os.environ["ISING_TAIL"] = f"{tail},{h2}" if tail else "0"
                with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
                    s.init(); s.sweep_timed(max(8, sweeps // 4))
                    best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(2))
                res.append((best, H, tail, h2))
    res.sort(reverse=True)
    print(f"{Y} rows x {X} cols: " + "  ".join(f"H={h} tail={t},{k}: {v:.0f}" for v, h, t, k in res[:6]) + f"   ...  no tail best: {max(v for v, h, t, k in res if t == 0):.0f}", flush=True)
