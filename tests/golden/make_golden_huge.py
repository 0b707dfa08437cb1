"""Oracle goldens for lattices that do not fit the dev container's memory in the reference's packed form: the pinned CPU
oracle STREAMED over the lattice in chunks of rows (tests/test_gpu_fullsize.py: 2^20 columns x 2^17 rows = 2^37 spins =
64 GiB at 4 bit per spin).  A chunk is a slab of a ring with 16 ghost rows on either side (oracle.OracleGhostSlab, the CPU
counterpart of the product's ring slabs): its own rows and ghost rows are initialised from the seed (latticeInit_k draws
depend on the global row only, optimized/main.cu:92-151), `sweeps` sweeps run over slab and ghost rows -- the ghost rows
with the draws their owners make --, which leaves the chunk's rows and one row beyond exact; counts and the black-site bond
sum of the chunk's rows are added up.  Numbers only; ~10 min on 8 cores.

Usage: python tests/golden/make_golden_huge.py [X Y seed sweeps chunk_rows]   -> tests/golden/huge_1048576x131072.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402
from oracle.pyoracle import _u64, lib  # noqa: E402


def main():
    X, Y, seed, sweeps, R = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (1 << 20, 1 << 17, 4321, 2, 4096)))
    G = 16
    assert Y % R == 0 and R % 16 == 0 and 2 * sweeps <= G - 2
    temp = oracle.CRIT_TEMP
    nchunks = Y // R
    pts = {0: [0, 0, 0], sweeps: [0, 0, 0]}  # sweeps -> [up, down, bond_equal]
    t0 = time.time()
    for k in range(nchunks):
        s = oracle.OracleGhostSlab(X, R, seed, temp, nchunks, k, G)
        # own rows and both ghost blocks from the seed, each block at its global row (around the lattice at the ends)
        for e0, rows, g0 in ((0, G, (k * R - G) % Y), (G, R, k * R), (G + R, G, ((k + 1) * R) % Y)):
            rc = lib().orc_init_slab(_u64(s.ext[0, e0:]), _u64(s.ext[1, e0:]), X, rows, g0, C.c_uint64(seed))
            assert rc == 0
        for pt in (0, sweeps):
            if pt:
                s.sweep_ghost(1, sweeps)
            up, dw = C.c_uint64(), C.c_uint64()
            lib().orc_count(_u64(s.ext[0, G:]), _u64(s.ext[1, G:]), X, R, C.byref(up), C.byref(dw))
            a = lib().orc_bond_equal_slab(_u64(s.ext[0, G:]), _u64(s.ext[1, G:]), _u64(s.ext[1, G - 1]), _u64(s.ext[1, G + R]), X, R, k * R)
            pts[pt][0] += int(up.value)
            pts[pt][1] += int(dw.value)
            pts[pt][2] += int(a)
        print(f"chunk {k + 1}/{nchunks} [{time.time() - t0:.0f} s] running totals {pts}", flush=True)
        del s
    t = np.float32(temp)
    out = {"generated_by": "tests/golden/make_golden_huge.py (pinned oracle, streamed in chunks of rows with 16 ghost rows)",
           "X": X, "Ytot": Y, "seed": seed, "temp": float(t), "temp_bits": int(t.view(np.uint32)), "chunk_rows": R,
           "points": [{"sweeps": p, "up": v[0], "down": v[1], "bond_equal": v[2]} for p, v in sorted(pts.items())]}
    path = os.path.join(HERE, f"huge_{X}x{Y}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, out["points"])


if __name__ == "__main__":
    main()
