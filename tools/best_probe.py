#!/usr/bin/env python3
"""Fused launches at the best (width, height, grid) found per lattice: A/B of library variants (ISING_LIB)."""
import os, sys, subprocess
sys.path.insert(0, __file__.rsplit("/", 2)[0])
CASES = ((8192, 8192, 1, 1, 768), (8192, 8192, 0, 2, 768), (16384, 8192, 1, 2, 512), (16384, 16384, 0, 4, 1024), (16384, 16384, 1, 2, 768),
         (32768, 32768, 0, 8, 1536), (65536, 65536, 0, 8, 1536))
if len(sys.argv) > 1:
    import ising_gpu_amd as ig
    X, Y, wide, H = map(int, sys.argv[1:5])
    os.environ["ISING_FUSED"] = "1"; os.environ["ISING_FUSED_WIDE"] = str(wide)
    sweeps = max(64, min(4096, (1 << 33) // (X * Y) * 8))
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        s.init(); s.sweep_timed(max(8, sweeps // 8))
        best = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(3))
    print(f"{Y}x{X} w{wide} H{H} g{os.environ.get('ISING_FUSED_WGS')}: {best:6.1f}", end=" | ", flush=True)
else:
    for c in CASES:
        subprocess.run([sys.executable, __file__, *map(str, c[:4])], env=dict(os.environ, ISING_FUSED_WGS=str(c[4])), stderr=subprocess.DEVNULL)
    print()
