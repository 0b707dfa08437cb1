#!/usr/bin/env python3
"""(Round 2 form of this probe -- the one that set the 1.5 * 2^24 threshold; round 4 rewrote tools/small_fused_probe.py.)  Lattices below 2^26 spins: what AUTO picks (dense layout, one launch per colour) against fused launches on the ballot layout
(one- and two-row units, ticket counters, grid).  Usage: small_fused_probe.py [X Y ...]"""
import os, sys, subprocess
sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y = map(int, sys.argv[2:4])
    sweeps = 4096
    with ig.IsingSlab(16384, 16384, seed=1, temp=ig.CRIT_TEMP_F32) as s:
        s.init(); s.sweep_timed(512)
    def run(**kw):
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, **kw) as s:
            s.init(); s.sweep_timed(64)
            return max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3)), s.current_layout(), s.strip_rows
    v, lay, H = run()
    out = [f"auto[layout {lay}, H={H}] {v:7.1f}"]
    os.environ["ISING_FUSED"] = "1"
    for H in (1, 2):
        v, _, _ = run(layout=ig.LAYOUT_BALLOT, strip_rows=H)
        out.append(f"fused H={H} {v:7.1f}")
    print(f"{Y:6d} x {X:6d} t2={os.environ.get('ISING_FUSED_TICKETS2', '-')} wgs={os.environ.get('ISING_FUSED_WGS', 'auto'):>5s}  " + "  ".join(out), flush=True)
else:
    sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(8192, 4096), (16384, 2048), (8192, 2048), (8192, 6144)]
    for X, Y in sizes:
        for t2 in ("2", "4"):
            for g in ("512", "768", "1024"):
                subprocess.run([sys.executable, __file__, "case", str(X), str(Y)], env=dict(os.environ, ISING_FUSED_TICKETS2=t2, ISING_FUSED_WGS=g), stderr=subprocess.DEVNULL)
