#!/usr/bin/env python3
"""Fused launches: throughput against strip height and size of the persistent grid (ISING_FUSED_WGS).  (Rounds 2-3 also varied the
workgroup width; 8-wave workgroups are gone, the "wide" column is kept at 0.)
usage: grid_probe.py [X Y]...   (one process per lattice; the grid cap is read once per process, so one subprocess per cap)"""
import os, sys, subprocess
sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y, wide = map(int, sys.argv[2:5])
    os.environ["ISING_FUSED"] = "1"
    sweeps = max(64, min(4096, (1 << 33) // (X * Y) * 8))
    out = []
    for H in map(int, sys.argv[5:]):
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
            s.init(); s.sweep_timed(max(8, sweeps // 8))
            out.append(f"H={H}: {max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(2)):7.1f}")
    print(f"{Y:6d} x {X:6d} wide={wide} wgs={os.environ.get('ISING_FUSED_WGS', 'auto'):>5s}  " + "  ".join(out), flush=True)
else:
    sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(8192, 8192), (16384, 16384)]
    for X, Y in sizes:
        for wide, grids in ((0, (768, 1024, 1280, 1536)),):
            for g in grids:
                subprocess.run([sys.executable, __file__, "case", str(X), str(Y), str(wide), "1", "2", "4", "8"], env=dict(os.environ, ISING_FUSED_WGS=str(g)),
                               stderr=subprocess.DEVNULL)
