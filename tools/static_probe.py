#!/usr/bin/env python3
"""(Round 4; needs the library at commit 851dc8e: the STATIC form was measured, lost and was removed -- profiles/static_probe_r04a/b.txt.)
Fused launches without tickets (ISING_FUSED_STATIC=1: one resident workgroup per unit of a level, ising_ballot.hip STATIC) against the
library's default, by strip height, on lattices whose level fits the chip.  Every case also checks counts and bond sum against the
default path after the same sweeps.  Usage: static_probe.py [X Y ...]  -> flips/ns"""
import os
import subprocess
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sweeps = max(512, (1 << 36) // (X * Y) // 32 * 32)
    kw = dict(layout=ig.LAYOUT_BALLOT, strip_rows=H) if H else {}
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, **kw) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
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
        print("RESULT", best, s.strip_rows, s.current_layout(), int(s.fused), chk[0][0], chk[1])
    sys.exit(0)

sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(8192, 8192), (16384, 16384), (16384, 8192), (8192, 4096), (8192, 2048), (4096, 4096), (2048, 2048)]


def case(X, Y, H, **env):
    e = dict(os.environ)
    for k in ("ISING_FUSED_STATIC", "ISING_FUSED", "ISING_FUSED_WGS"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, __file__, "case", str(X), str(Y), str(H)], env=e, capture_output=True, text=True, timeout=600)
    res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if not res:
        return None, (r.stderr.strip().splitlines() or ["?"])[-1][:100]
    f = res[-1].split()
    return float(f[1]), (int(f[2]), int(f[3]), int(f[4]), f[5], f[6])


for X, Y in sizes:
    v0, d0 = case(X, Y, 0)
    print(f"{Y} x {X}: default {v0:7.1f} flips/ns (H = {d0[0]}, layout {d0[1]}, fused {d0[2]})", flush=True)
    for H in (1, 2, 4, 8, 16):
        if Y % H:
            continue
        nwc = (X // 2048 + 3) // 4
        nwg = (4 * nwc * (Y // H) + 15) // 16
        if nwg > 24 * 256 or nwg < 256:
            continue
        vt, dt = case(X, Y, H, ISING_FUSED="1")
        row = [f"tickets {vt if vt else float('nan'):7.1f}"]
        for per_cu in (2, 3, 4, 5, 6):
            if per_cu * 256 > nwg and (per_cu - 1) * 256 >= nwg:
                continue
            vs, ds = case(X, Y, H, ISING_FUSED="1", ISING_FUSED_STATIC="1", ISING_FUSED_WGS=str(per_cu * 256))
            ok = vs is not None and ds[3:] == d0[3:]
            row.append(f"static@{per_cu}: {vs if vs else float('nan'):7.1f}" + ("" if ok else " COUNTS DIFFER"))
        print(f"    H = {H:2d} ({nwg:5d} units a level): " + "   ".join(row), flush=True)
