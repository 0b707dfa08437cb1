#!/usr/bin/env python3
"""Where a fused launch's workgroup time goes, at the shapes ising_create picks (needs a -DISING_FUSED_TRACE variant build:
make -C ising_gpu_amd/csrc variant NAME=trace DEFS=-DISING_FUSED_TRACE; ISING_LIB=.../libising_hip_trace.so).
The library prints the trace when a slab is destroyed."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig
sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [(8192, 8192), (16384, 8192), (16384, 16384), (32768, 32768), (65536, 65536)]
for X, Y in sizes:
    sweeps = max(64, min(8192, (1 << 34) // (X * Y) * 8))
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        s.sweep_timed(sweeps)
        ms = s.sweep_timed(sweeps)
        print(f"{Y} x {X}: layout {s.layout}, H={s.strip_rows}, fused={int(s.fused)}: {X * Y * sweeps / (ms * 1e6):7.1f} flips/ns (trace build), {ms * 1e3 / (2 * sweeps):8.2f} us per colour", flush=True)
