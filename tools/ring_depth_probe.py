#!/usr/bin/env python3
"""Ring slabs against the lone slab's rate: a ring of ONE slab over the peer (IPC) transport -- everything a rank of an N-rank ring executes
except a real link -- by ghost depth G (= 2 x the sweeps a launch carries), launch form (fused / split) and strip height, next to the same rows as a lone slab.
Counts after the timed sweeps are compared with the lone slab's (a shape or a depth never changes results).
Usage: ring_depth_probe.py [--sweeps N] [--cases "G:form:H[:E],..."] X Y [X Y ...]      form: f = fused, s = split, a = the library's choice; H = 0: the library's
choice; E = exchange epochs per launch (ISING_RING_EPOCHS; absent or 0: the library's choice, 1: a launch per exchange)"""
import argparse
import os
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402,F401
import ising_gpu_amd as ig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=0)
ap.add_argument("--cases", default="64:f:0:1,64:f:0:0,32:f:0:0,128:f:0:0,64:f:16:0,64:f:0:4")
ap.add_argument("--transport", default="ipc")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("sizes", nargs="*", type=int)
args = ap.parse_args()
sizes = [tuple(args.sizes[i:i + 2]) for i in range(0, len(args.sizes), 2)] or [(65536, 8192), (65536, 16384)]


def rate(sweep, sync, X, Y, n):
    sweep(min(n, 256))
    sync()
    best = 0.0
    for _ in range(args.reps):
        t0 = time.perf_counter()
        sweep(n)
        sync()
        best = max(best, X * Y * n / (time.perf_counter() - t0) * 1e-9)
    return best


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


for X, Y in sizes:
    n = args.sweeps or max(256, (1 << 37) // (X * Y) // 256 * 256)
    total = min(n, 256) + args.reps * n
    ref = None
    for form in ("f", "s"):
        setenv(ISING_SPLIT="1" if form == "s" else "0", ISING_RING_GHOST=None)
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
            s.init()
            r = rate(s.sweep, s.synchronize, X, Y, n)
            cnt = s.count()
            assert ref is None or ref == cnt, (ref, cnt)
            ref = cnt
            print(f"{Y} x {X} lone, {'split' if form == 's' else 'fused'}: {r:7.1f} flips/ns  shape (H, wg/CU, lead) {s.launch_shape()}  {n} sweeps a call", flush=True)
    setenv(ISING_SPLIT=None)
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        r = rate(s.sweep, s.synchronize, X, Y, n)
        assert s.count() == ref
        print(f"{Y} x {X} lone, library's choice: {r:7.1f} flips/ns  shape {s.launch_shape()} split {s.split}", flush=True)
    for case in args.cases.split(","):
        G, form, H, E = (case.split(":") + ["0"])[:4]
        setenv(ISING_RING_GHOST=G if int(G) > 0 else None, ISING_SPLIT={"f": "0", "s": "1", "a": None}[form], ISING_RING_EPOCHS=E if int(E) > 0 else None)
        try:
            slab = ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, ring_halo=True, strip_rows=int(H))
        except ig.IsingError as e:
            print(f"{Y} x {X} ring of one G={G} {form} H={H}: {e}", flush=True)
            continue
        try:
            ring = ig.NativeRing(slab, transport=args.transport).init()
            r = rate(ring.sweep, ring.quiesce, X, Y, n)
            cnt = ring.count()
            print(f"{Y} x {X} ring of one ({args.transport}) G={G:>4} {form} H={H:>2} E={E:>2}: {r:7.1f} flips/ns  shape {slab.launch_shape()} split {slab.split}  "
                  f"{slab.max_sweeps_per_launch} sweeps a launch  counts {'==' if cnt == ref else '!='} lone", flush=True)
            ring.close()
        finally:
            slab.close()
    setenv(ISING_RING_GHOST=None, ISING_SPLIT=None, ISING_RING_EPOCHS=None)
