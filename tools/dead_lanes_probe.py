"""Lattices whose last wave column is partly dead (X not a multiple of 8192): layout AUTO against the ballot and the dense layout.  profiles/dead_lanes_probe_r04.txt"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import ising_gpu_amd as ig
def rate(X, Y, layout):
    sweeps = max(256, (1 << 36) // (X * Y) // 32 * 32)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=layout) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            s.sweep(64); s.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter(); s.sweep(sweeps); s.synchronize()
            best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
        return best, s.strip_rows
for X, Y in [(20480, 4096), (20480, 8192), (20480, 16384), (12288, 8192), (12288, 16384), (6144, 16384), (6144, 32768), (28672, 4096), (28672, 8192), (28672, 28672), (20480, 20480), (12288, 12288)]:
    a = rate(X, Y, ig.LAYOUT_AUTO); b = rate(X, Y, ig.LAYOUT_BALLOT); d = rate(X, Y, ig.LAYOUT_DENSE)
    gx = X // 2048; nwc = (X + 8191) // 8192
    print(f"{Y} x {X} ({X*Y/2**26:.2f} x 2^26 spins, {gx}/{4*nwc} of the lanes alive): library {a[0]:6.0f}; ballot {b[0]:6.0f} (H = {b[1]}); dense {d[0]:6.0f}", flush=True)
