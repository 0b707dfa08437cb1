#!/usr/bin/env python3
"""Strip height x persistent grid for single mid-size lattices at HEAD (fused launches): midsize_probe.py [X Y] -> flips/ns."""
import os
import subprocess
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 3:
    import ising_gpu_amd as ig
    X, Y, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sweeps = max(256, (1 << 37) // (X * Y) // 8)
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, strip_rows=H) as s:
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
        print("RESULT", best, s.strip_rows)
    sys.exit(0)
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 16384)
print(f"{Y}x{X}: rows = strip height, columns = workgroups per CU (0 = the library's choice)")
for H in (0, 2, 4, 8, 16):
    row = []
    for per_cu in (0, 3, 4, 5, 6):
        env = dict(os.environ)
        env.pop("ISING_FUSED_WGS", None)
        if per_cu:
            env["ISING_FUSED_WGS"] = str(256 * per_cu)
        r = subprocess.run([sys.executable, __file__, str(X), str(Y), str(H)], env=env, capture_output=True, text=True)
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        row.append(f"{float(res[-1].split()[1]):7.1f}" if res else " FAILED")
    print(f"H = {H:2d}: " + "  ".join(row), flush=True)
