#!/usr/bin/env python3
"""(Round 4; needs the library at commit 9f35ffa: the sweep graphs were measured -- no gain -- and removed; profiles/sweep_graph_probe_r04.txt.)
Lattices under 1.5 * 2^24 spins (the dense layout, one launch per colour): sweeps replayed from a captured hipGraph (default) against
one launch per colour from the host (ISING_SWEEP_GRAPH=0).  Usage: small_probe.py [X Y ...] -> flips/ns, us per sweep"""
import os
import subprocess
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y = int(sys.argv[2]), int(sys.argv[3])
    sweeps = 4096
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            s.sweep(256)
            s.synchronize()
        s.init().sweep(96)
        chk = (s.count(), s.bond_equal())
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            s.sweep(sweeps)
            s.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("RESULT", X * Y * sweeps / best * 1e-9, best / sweeps * 1e6, s.current_layout(), chk[0][0], chk[1])
    sys.exit(0)
sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(2048, 2048), (4096, 4096), (8192, 2048), (2048, 8192), (4096, 2048), (6144, 4096), (8192, 3072)]
for X, Y in sizes:
    out = []
    for g in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "case", str(X), str(Y)], env=dict(os.environ, ISING_SWEEP_GRAPH=g), capture_output=True, text=True, timeout=600)
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        out.append(res[-1].split()[1:] if res else None)
    if None in out:
        print(f"{Y} x {X}: FAILED {out}", flush=True)
        continue
    same = "same counts" if out[0][3:] == out[1][3:] else "COUNTS DIFFER"
    print(f"{Y:5d} x {X:5d} (layout {out[0][2]}): graph replay {float(out[0][0]):7.1f} flips/ns ({float(out[0][1]):5.2f} us per sweep)   one launch per colour from the host "
          f"{float(out[1][0]):7.1f} ({float(out[1][1]):5.2f} us)   x {float(out[0][0]) / float(out[1][0]):.2f}   {same}", flush=True)
