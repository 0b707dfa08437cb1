#!/usr/bin/env python3
"""Lone slabs around 2^24 spins: what the library picks (layout AUTO) against fused launches of one-row units on the ballot layout at 2 .. 5 workgroups per CU and
against the dense layout (tile launches / one launch per colour).  profiles/small_fused_probe_r04.txt
Usage: small_fused_probe.py [X Y ...]"""
import os
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
import ising_gpu_amd as ig  # noqa: E402

os.environ["ISING_ABORT_POLLS"] = "40000"


def rate(X, Y, layout, env, H=0):
    for k in ("ISING_FUSED", "ISING_TILES", "ISING_FUSED_WGS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sweeps = max(512, (1 << 35) // (X * Y) // 64 * 64)
    try:
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=layout, strip_rows=H) as s:
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
            return best, f"{'ballot' if s.layout == ig.LAYOUT_BALLOT else 'dense'}{' fused H=%d' % s.strip_rows if s.fused else ''}{' tiles' if s.tiled else ''}"
    except ig.IsingError:
        return -1.0, "-"


sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or (
    [(8192, y) for y in (1024, 1280, 1536, 1792, 2048, 2560, 3072, 4096)] + [(16384, y) for y in (640, 768, 1024, 1152, 1408, 1536, 2048)] +
    [(24576, y) for y in (512, 640, 768, 896, 1024)] + [(32768, y) for y in (384, 512, 640, 768, 1024, 2048)] + [(65536, y) for y in (256, 320, 384, 512, 1024)] +
    [(12288, y) for y in (1024, 1536, 2048, 4096)] + [(6144, y) for y in (2048, 3072, 4096, 8192)] + [(20480, y) for y in (1024, 2048)])
for X, Y in sizes:
    nwc = (X + 8191) // 8192
    lib, how = rate(X, Y, ig.LAYOUT_AUTO, {})
    out = []
    for wgs in (2, 3, 4, 5):
        out.append(f"{wgs}: {rate(X, Y, ig.LAYOUT_BALLOT, {'ISING_FUSED': '1', 'ISING_FUSED_WGS': str(256 * wgs)}, 1)[0]:5.0f}")
    d = rate(X, Y, ig.LAYOUT_DENSE, {})
    print(f"{Y} x {X} ({X * Y / 2**24:.2f} x 2^24 spins, {nwc * Y // 4} tickets a level of one-row units): library {lib:6.0f} ({how}); ballot, fused, H = 1, per CU "
          + "  ".join(out) + f"; dense {d[0]:6.0f} ({d[1]})", flush=True)
