#!/usr/bin/env python3
"""What a fused launch's fixed cost (~60 us: staggered start + uneven tail) means by lattice size: sweeps per launch 32 (round 2)
against ~50 ms worth (round 3's default).  launch_len_probe.py -> flips/ns per size and cap."""
import os
import subprocess
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

sys.path.insert(0, __file__.rsplit("/", 2)[0])

if len(sys.argv) > 1:
    import ising_gpu_amd as ig
    X, Y, sweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            s.sweep(64)
            s.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            s.sweep(sweeps)
            s.synchronize()
            best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
        print("RESULT", best, s.max_sweeps_per_launch, s.strip_rows)
    sys.exit(0)

for X, Y, sweeps in ((8192, 4096, 8192), (8192, 8192, 4096), (16384, 8192, 4096), (16384, 16384, 2048), (32768, 32768, 512), (65536, 65536, 128)):
    row = []
    for cap in ("32", None, "128", "1024"):
        env = dict(os.environ)
        env.pop("ISING_FUSED_MAX_SWEEPS", None)
        if cap:
            env["ISING_FUSED_MAX_SWEEPS"] = cap
        r = subprocess.run([sys.executable, __file__, str(X), str(Y), str(sweeps)], env=env, capture_output=True, text=True)
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        row.append((cap or "auto", *res[-1].split()[1:]) if res else (cap, "FAILED", r.stderr[-200:]))
    print(f"{Y}x{X}: " + "   ".join(f"cap {c[0]:>4s} ({c[2]} per launch): {float(c[1]):7.1f}" for c in row if c[1] != "FAILED"), flush=True)
