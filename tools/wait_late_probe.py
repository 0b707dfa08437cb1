#!/usr/bin/env python3
"""A/B: fused launches whose units draw their first two rows (ISING_FUSED_WAIT_LATE=2, the default) or their first row (=1) before they wait for
their parents against units that wait first (=0), by strip height and workgroups per CU.  (profiles/ticket_early_probe_r04.txt was made with this
script too, at a commit that could request the next ticket a row early: ISING_FUSED_TICKET_EARLY.)
Usage: wait_late_probe.py [X Y ...] -> flips/ns"""
import os
import subprocess
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sweeps = max(256, (1 << 37) // (X * Y) // 32 * 32)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, strip_rows=H) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            s.sweep(64)
            s.synchronize()
        s.init().sweep(96)
        chk = (s.count(), s.bond_equal())
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            s.sweep(sweeps)
            s.synchronize()
            best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
        print("RESULT", best, s.strip_rows, chk[0][0], chk[1])
    sys.exit(0)

sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(8192, 4096), (8192, 8192), (16384, 8192), (16384, 16384), (24576, 24576), (32768, 32768), (65536, 65536)]
libs = {"late2": "2", "late1": "1", "early": "0"}  # ISING_FUSED_WAIT_LATE: 0 wait first; 1 behind the first draw phase; 2 (default) behind the second too
for X, Y in sizes:
    print(f"{Y} x {X}: rows = strip height (0 = the library's choice), columns = workgroups per CU (0 = the library's choice); late2 / late1 / early", flush=True)
    ref = None
    for H in (0, 1, 2, 4, 8, 16):
        if H and (Y % H or (X * Y >= (1 << 30) and H < 8) or (X * Y <= (1 << 28) and H > 8) or (X * Y <= (1 << 26) and H > 4) or (X * Y > (1 << 26) and H < 2)):
            continue
        row = []
        for per_cu in (0, 3, 4, 5, 6):
            cell = []
            for name, lib in libs.items():
                env = dict(os.environ, ISING_FUSED_WAIT_LATE=lib, ISING_ABORT_POLLS="40000")
                env.pop("ISING_FUSED_WGS", None)
                if per_cu:
                    env["ISING_FUSED_WGS"] = str(256 * per_cu)
                r = subprocess.run([sys.executable, __file__, "case", str(X), str(Y), str(H)], env=env, capture_output=True, text=True, timeout=900)
                res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
                if not res:
                    cell.append(" FAILED")
                    continue
                f = res[-1].split()
                ref = ref or f[3:]
                cell.append(f"{float(f[1]):7.1f}" + ("" if f[3:] == ref else "!"))
            row.append("/".join(cell))
        print(f"  H = {H:2d}: " + "   ".join(row), flush=True)
