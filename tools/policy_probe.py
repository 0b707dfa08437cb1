#!/usr/bin/env python3
"""What ising_create picks (layout AUTO) against the alternatives, per lattice: flips/ns after a preheat, best of 2.
usage: policy_probe.py [X Y]..."""
import os, sys, subprocess
sys.path.insert(0, __file__.rsplit("/", 2)[0])
if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y = map(int, sys.argv[2:4])
    sweeps = max(32, min(4096, (1 << 35) // (X * Y) * 8)) // 32 * 32
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:  # preheat: the clock ramp takes ~40 ms under load
        s.init(); s.sweep_timed(2 * sweeps)
    out = []
    for name, lay, env in (("auto", ig.LAYOUT_AUTO, {}), ("ballot, one launch per colour", ig.LAYOUT_BALLOT, {"ISING_FUSED": "0"}), ("dense", ig.LAYOUT_DENSE, {})):
        for k in ("ISING_FUSED",):
            os.environ.pop(k, None)
        os.environ.update(env)
        with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, layout=lay) as s:
            s.init(); s.sweep_timed(32)
            v = max(X * Y * sweeps / (s.sweep_timed(sweeps) * 1e6) for _ in range(2))
            out.append(f"{name} [layout {s.layout}, H={s.strip_rows}, fused={int(s.fused)}]: {v:7.1f}")
    print(f"{Y:6d} x {X:6d}  " + "   ".join(out), flush=True)
else:
    sizes = [tuple(map(int, sys.argv[i:i + 2])) for i in range(1, len(sys.argv), 2)] or [
        (8192, 4096), (8192, 8192), (12288, 8192), (16384, 8192), (8192, 16384), (16384, 16384), (24576, 24576), (16384, 32768), (32768, 32768),
        (65536, 16384), (65536, 32768), (65536, 65536), (131072, 16384), (131072, 131072)]
    for X, Y in sizes:
        subprocess.run([sys.executable, __file__, "case", str(X), str(Y)], stderr=subprocess.DEVNULL)
