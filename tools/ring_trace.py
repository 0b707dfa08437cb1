#!/usr/bin/env python3
"""One GPU of a ring under rocprofv3 --kernel-trace: a ring of one slab (ring_halo) sweeping through the library's ring
schedule.  Usage: rocprofv3 --kernel-trace -d DIR -o trace -- python tools/ring_trace.py [rccl|copy] [X Y sweeps]
Then: python tools/ring_trace.py --analyze DIR  (gaps between consecutive interior launches, where the edge-row launch sits)
      python tools/ring_trace.py --window DIR   (the kernels of two colours in time order)."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])

if len(sys.argv) > 2 and sys.argv[1] == "--window":  # every kernel between the 40th and the 42nd interior launch, in time order
    db = glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True)[0]
    rows = sqlite3.connect(db).execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
    big = [r for r in rows if "update_k" in r[0] and r[3] > 100000]
    t0, t1 = big[40][1], big[42][2]
    for r in rows:
        if t0 <= r[1] <= t1:
            print(f"{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:7.1f} us  grid {r[3]:8d}  stream {r[4]}  {r[0][:64]}")
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    db = glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, stream_id from kernels order by start").fetchall()
    big = [r for r in rows if "update_k" in r[0] and r[3] > 100000]
    small = [r for r in rows if "update_k" in r[0] and r[3] <= 100000]
    gaps = [(b[1] - a[2]) / 1e3 for a, b in zip(big, big[1:])]
    durs = [(r[2] - r[1]) / 1e3 for r in big]
    sd = [(r[2] - r[1]) / 1e3 for r in small]
    n = len(gaps)
    gs = sorted(gaps)
    print(f"{len(big)} interior launches: mean {sum(durs) / len(durs):.1f} us; gaps between consecutive ones: median {gs[n // 2]:.1f} us, mean {sum(gaps) / n:.1f}, p90 {gs[int(0.9 * n)]:.1f}")
    if sd:
        print(f"{len(small)} edge-row launches: mean {sum(sd) / len(sd):.1f} us, max {max(sd):.1f}")
    other = {}
    for r in rows:
        if "update_k" not in r[0]:
            other.setdefault(r[0][:40], []).append((r[2] - r[1]) / 1e3)
    for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1]))[:5]:
        print(f"   {k:40s} x{len(v):5d} mean {sum(v) / len(v):8.1f} us")
    sys.exit(0)

import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
X, Y, sweeps = (int(v) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (65536, 65536, 48)))
os.environ["ISING_FUSED"] = "0"
if mode == "single":
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32) as s:
        s.init().sweep(sweeps)
        s.synchronize()
elif mode == "rccl":
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
        ring = ig.NativeRing(s).init()
        ring.sweep(sweeps)
        ring.quiesce()
        ring.close()
else:
    os.environ["ISING_RING_INLINE"] = "1" if mode == "inline" else "0"
    with ig.IsingSlab(X, Y, seed=1, temp=ig.CRIT_TEMP_F32, ring_halo=True) as s:
        ring = ig.SlabSet([s]).init()
        ring.sweep(sweeps)
        ring.synchronize()
