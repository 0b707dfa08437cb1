#!/usr/bin/env python3
"""Larger lattices: fused launches (4-wave workgroups) against strip height and grid, and the plain-launch default."""
import os, sys, subprocess
sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y, fused = map(int, sys.argv[2:5])
    os.environ["ISING_FUSED"] = str(fused)
    sweeps = max(32, min(4096, (1 << 35) // (X * Y) * 8)) // 32 * 32  # ~0.2 s per measurement
    out = []
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT) as s:  # preheat: the clock ramp takes ~40 ms under load
        s.init(); s.sweep_timed(2 * sweeps)
    for H in map(int, sys.argv[5:]):
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
            s.init(); s.sweep_timed(32)
            out.append(f"H={s.strip_rows}: {max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(2)):7.1f}")
    print(f"{Y:6d} x {X:6d} {'fused' if fused else 'plain'} wgs={os.environ.get('ISING_FUSED_WGS', 'auto'):>5s}  " + "  ".join(out), flush=True)
else:
    sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)]
    for X, Y in sizes:
        subprocess.run([sys.executable, __file__, "case", str(X), str(Y), "0", "0"], stderr=subprocess.DEVNULL)
        for g in (1024, 1280, 1536):
            subprocess.run([sys.executable, __file__, "case", str(X), str(Y), "1", "2", "4", "8", "16"], env=dict(os.environ, ISING_FUSED_WGS=str(g)), stderr=subprocess.DEVNULL)
