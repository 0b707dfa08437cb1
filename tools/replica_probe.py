#!/usr/bin/env python3
"""Independent lattices side by side on one GPU (each slab has its own stream): aggregate flips/ns for K replicas of a small
lattice, against one replica alone.  usage: replica_probe.py [X Y]"""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
import torch  # (only for stream handles; the library shares torch's HIP runtime, ising_gpu_amd/_lib.py)
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 8192)
sweeps = max(64, min(4096, (1 << 34) // (X * Y) * 8)) // 32 * 32
for K in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream() for k in range(K)]
    slabs = [ig.IsingSlab(X, Y, seed=1234 + k, temp=ig.CRIT_TEMP_F32) for k in range(K)]
    for s, st in zip(slabs, streams):
        s.set_stream(st.cuda_stream)
        s.init()
    try:
        for s in slabs:
            s.sweep(sweeps)
        for s in slabs:
            s.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            for j in range(0, sweeps, 32):        # interleave the launches of the replicas, 32 sweeps (one fused launch) at a time
                for s in slabs:
                    s.sweep(32)
            for s in slabs:
                s.synchronize()
            dt = time.perf_counter() - t0
            best = max(best, K * X * Y * sweeps / dt / 1e9)
        print(f"{Y} x {X}: {K} replica(s) [layout {slabs[0].layout}, H={slabs[0].strip_rows}, fused={int(slabs[0].fused)}]: {best:7.1f} flips/ns aggregate", flush=True)
    finally:
        for s in slabs:
            s.close()
