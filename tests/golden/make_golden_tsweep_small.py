"""Golden series for batches of small lattices (round 6: cuIsing --tsweep on the quad path), from the CPU oracle; outputs are data only.

  tsweep_2048.json -- 2048x2048, seeds 1234 and 1235 (--tsweep-chains 2), T = 1.50 .. 3.00 step 0.05 (31 values): (up, down, bond-equal) after 0, 8, 16, 24, 32 sweeps
  tsweep_4096.json -- 4096x4096, seed 1234, the same temperatures: after 0, 8, 16 sweeps
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def series(X, Y, seeds, sweeps):
    temps = [round(1.5 + 0.05 * k, 2) for k in range(31)]
    out = {"X": X, "Y": Y, "seeds": list(seeds), "series": []}
    for seed in seeds:
        for t in temps:
            L = oracle.OracleLattice(X, Y, seed=seed, temp=t).init()
            pts = []
            for s in sweeps:
                L.sweep(s - L.it)
                up, dw = L.count()
                pts.append({"sweeps": s, "up": up, "down": dw, "bond_equal": L.bond_equal()})
            out["series"].append({"seed": seed, "temp": float(np.float32(t)), "temp_bits": int(np.float32(t).view(np.uint32)), "points": pts})
            print(X, "seed", seed, "T", t, pts[-1], flush=True)
    return out


if __name__ == "__main__":
    json.dump(series(2048, 2048, (1234, 1235), (0, 8, 16, 24, 32)), open(os.path.join(HERE, "tsweep_2048.json"), "w"), indent=1)
    json.dump(series(4096, 4096, (1234,), (0, 8, 16)), open(os.path.join(HERE, "tsweep_4096.json"), "w"), indent=1)
    print("golden fixtures written")
