import sys, time, os
sys.path.insert(0, '.')
import torch
import ising_gpu_amd as ig
tr, X, Y = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = ig.IsingSlab(X, Y, seed=99, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, ring_halo=True)
ring = ig.NativeRing(s, transport=tr).init()
print("strip rows", s.strip_rows, flush=True)
t0 = time.time(); n = 0
pat = (97, 32, 5, 64, 1, 33)
try:
    for i in range(600):
        m = pat[i % 6] if len(sys.argv) < 5 else int(sys.argv[4])
        ring.sweep(m); n += m
        if i % 20 == 19:
            c = ring.count(); print(n, "sweeps", c, f"{time.time()-t0:.1f}s", flush=True)
    print("OK", n)
except Exception as e:
    print("FAILED after", n, "sweeps, call", i, "len", m, str(e)[:80], flush=True)
