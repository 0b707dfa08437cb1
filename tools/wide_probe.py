#!/usr/bin/env python3
"""Wide-and-short lattices on the ballot layout (rows of 4 .. 16 wave columns, 2^26 .. 2^28 spins): the library's launch shape against strip heights 1 .. 16 x 3 .. 6
workgroups per CU.  profiles/wide_probe_r04.txt      Usage: wide_probe.py [X Y ...]"""
import os
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

os.environ["ISING_ABORT_POLLS"] = "40000"


def rate(X, Y, H, wgs):
    os.environ.pop("ISING_FUSED_WGS", None)
    if wgs:
        os.environ["ISING_FUSED_WGS"] = str(256 * wgs)
    sweeps = max(256, (1 << 36) // (X * Y) // 32 * 32)
    try:
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, strip_rows=H) as s:
            s.init()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                s.sweep(64)
                s.synchronize()
            best = 0
            for _ in range(3):
                t0 = time.perf_counter()
                s.sweep(sweeps)
                s.synchronize()
                best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
            return best, s.strip_rows
    except ig.IsingError:
        return -1.0, 0


LIB_ONLY = len(sys.argv) > 1 and sys.argv[1] == "--library-only"
if LIB_ONLY:
    del sys.argv[1]
sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(65536, 1024), (65536, 2048), (65536, 4096), (32768, 2048), (32768, 4096), (131072, 1024), (131072, 2048), (24576, 4096), (16384, 4096)]
for X, Y in sizes:
    lib, h = rate(X, Y, 0, 0)
    print(f"{Y} x {X} ({X * Y / 2**26:.2f} x 2^26 spins, {(X + 8191) // 8192} wave columns a row): library {lib:6.0f} (H = {h})", flush=True)
    for H in (() if LIB_ONLY else (1, 2, 4, 8, 16)):
        if Y % H or (X * Y) // H < (1 << 21):
            continue
        print(f"   H = {H:2d}: " + "  ".join(f"{w or 'lib'}: {rate(X, Y, H, w)[0]:5.0f}" for w in (0, 3, 4, 5, 6)), flush=True)
