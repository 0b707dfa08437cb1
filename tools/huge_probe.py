#!/usr/bin/env python3
"""Very large lattices (64-bit offsets everywhere?): counts and bond sums after a few sweeps must agree between the fused
ballot launches, one launch per colour and the dense layout; also the time per sweep.  usage: huge_probe.py [X Y ...]"""
import os, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ising_gpu_amd as ig  # noqa: E402

args = [int(v) for v in sys.argv[1:]] or [262144, 262144, 524288, 524288]
for X, Y in zip(args[::2], args[1::2]):
    res = {}
    for name, lay, env in (("fused", ig.LAYOUT_BALLOT, {"ISING_FUSED": "1"}), ("per colour", ig.LAYOUT_BALLOT, {"ISING_FUSED": "0"}), ("dense", ig.LAYOUT_DENSE, {})):
        os.environ.pop("ISING_FUSED", None)
        os.environ.update(env)
        with ig.IsingSlab(X, Y, seed=4321, temp=ig.CRIT_TEMP_F32, layout=lay) as s:
            s.init()
            c0 = s.count()
            ms = s.sweep_timed(3)
            # (the bond sum of a ballot slab works on a dense-order image of the whole slab: twice the memory)
            res[name] = (c0, s.count(), s.bond_equal() if X * Y < (1 << 40) else None)
            print(f"{Y} x {X} {name:10s} H={s.strip_rows:2d}: {X * Y * 3 / (ms * 1e6):7.1f} flips/ns  {res[name]}", flush=True)
    assert res["fused"] == res["per colour"] == res["dense"], "MISMATCH"
    print(f"{Y} x {X}: all three agree", flush=True)
