"""Generate the committed golden fixtures from the CPU oracle (run in the dev container; outputs are data only).

  small_states.json   -- full-state SHA-256 + counts + bond sums for small lattices after 0/1/2/17 sweeps
  tsweep_8192.json    -- BASELINE config 5: 8192x8192, seed 1234, T = 1.50 .. 3.00 step 0.05 (31 values),
                         (up, down, bond-equal) after 0, 4, 8, 12, 16 sweeps
  exp_tables.json     -- the ten FP32 exp-table entries (as uint32 bit patterns) per test temperature

The oracle itself is pinned against the reference's README transcripts by oracle/pin_readme_kats.py
(tests/golden/readme_kat_pin.json).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def state_hash(L):
    h = hashlib.sha256()
    h.update(L.black.tobytes())
    h.update(L.white.tobytes())
    return h.hexdigest()


def small_states():
    out = []
    for X, Y, seed, temp in [(2048, 16, oracle.SEED_DEF, 1.5), (2048, 16, 1234, oracle.CRIT_TEMP),
                             (4096, 64, 1234, oracle.CRIT_TEMP), (4096, 256, 7, 2.0), (6144, 48, 99, 3.0),
                             (8192, 128, oracle.SEED_DEF, 0.1 * oracle.CRIT_TEMP)]:
        L = oracle.OracleLattice(X, Y, seed=seed, temp=temp).init()
        rec = {"X": X, "Y": Y, "seed": seed, "temp_bits": int(np.float32(temp).view(np.uint32)), "temp": float(np.float32(temp)),
               "states": []}
        done = 0
        for upto in (0, 1, 2, 17):
            L.sweep(upto - done)
            done = upto
            up, dw = L.count()
            rec["states"].append({"sweeps": upto, "sha256": state_hash(L), "up": up, "down": dw, "bond_equal": L.bond_equal(),
                                  "black_row0_head": [f"{int(v):016x}" for v in L.black[0, :4]]})
        out.append(rec)
    return out


def tsweep():
    temps = [round(1.5 + 0.05 * k, 2) for k in range(31)]
    out = {"X": 8192, "Y": 8192, "seed": 1234, "series": []}
    for t in temps:
        L = oracle.OracleLattice(8192, 8192, seed=1234, temp=t).init()
        pts = []
        for s in (0, 4, 8, 12, 16):
            L.sweep(s - L.it)
            up, dw = L.count()
            pts.append({"sweeps": s, "up": up, "down": dw, "bond_equal": L.bond_equal()})
        out["series"].append({"temp": float(np.float32(t)), "temp_bits": int(np.float32(t).view(np.uint32)), "points": pts})
        print("T", t, pts[-1], flush=True)
    return out


def exp_tables():
    out = []
    for t in [1.5, 2.0, 2.26918, oracle.CRIT_TEMP, 3.0, 0.1 * oracle.CRIT_TEMP, 10.0]:
        tab = oracle.exp_table(t)
        out.append({"temp_bits": int(np.float32(t).view(np.uint32)), "temp": float(np.float32(t)),
                    "table_bits": [int(v) for v in tab.reshape(-1).view(np.uint32)]})
    return out


if __name__ == "__main__":
    json.dump(small_states(), open(os.path.join(HERE, "small_states.json"), "w"), indent=1)
    json.dump(exp_tables(), open(os.path.join(HERE, "exp_tables.json"), "w"), indent=1)
    json.dump(tsweep(), open(os.path.join(HERE, "tsweep_8192.json"), "w"), indent=1)
    print("golden fixtures written")
