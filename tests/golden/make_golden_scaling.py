"""Oracle goldens for every point of a scaling run of bench.py: counts after 0 / 5 / 25 / 144 sweeps
(the driver's --warmup 5 --steps 20 and the default 16 + 128) of every TOTAL lattice the three workloads produce at
N = 1, 2, 4, 8 ranks, T = CRIT_TEMP, seed 1234.  Results do not depend on the decomposition (optimized/main.cu:514: the
Philox stream id uses the global block row; :1590-1591: total = ndev*Y x X), so a record is keyed on the total lattice:

  config3 (weak)    65536 columns x 65536*N rows       N = 1: tests/golden/bench_65536_tc.json (not repeated here)
  config4 (weak)    131072 columns x 16384*N rows      N = 8 is BASELINE config 4, 131072^2
  strong            65536 x 65536 whole, 65536/N rows per rank: bench_65536_tc.json at every N

Numbers only, from the pinned CPU oracle (oracle/ising_oracle.c).  Resumable: a lattice whose record is complete in
scaling.json is skipped; the file is rewritten after every point.  About 2-3 h on 8 cores, up to 17 GiB of host memory.

Usage: python tests/golden/make_golden_scaling.py [name ...]     names: c4n1 c4n2 c4n4 c4n8 c3n2 c3n4 c3n8 (default: all)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

OUT = os.environ.get("ISING_GOLDEN_OUT", os.path.join(HERE, "scaling.json"))  # (ISING_GOLDEN_OUT: a run on another host, e.g. a GPU box's 16 cores under gpurun)
POINTS = (0, 5, 25, 144)
LATTICES = {  # name -> (X, Ytot), cheapest first
    "c4n1": (131072, 16384), "c4n2": (131072, 32768), "c3n2": (65536, 131072), "c4n4": (131072, 65536),
    "c4n8": (131072, 131072), "c3n4": (65536, 262144), "c3n8": (65536, 524288),
}


def load():
    try:
        return json.load(open(OUT))
    except (OSError, ValueError):
        t = np.float32(oracle.CRIT_TEMP)
        return {"generated_by": "tests/golden/make_golden_scaling.py (pinned CPU oracle; keyed on the TOTAL lattice, any decomposition)",
                "seed": 1234, "temp": float(t), "temp_bits": int(t.view(np.uint32)), "lattices": []}


def main():
    names = sys.argv[1:] or list(LATTICES)
    doc = load()
    for name in names:
        X, Y = LATTICES[name]
        rec = next((r for r in doc["lattices"] if r["X"] == X and r["Ytot"] == Y), None)
        if rec and {p["sweeps"] for p in rec["points"]} >= set(POINTS):
            print(f"{name}: complete, skipped", flush=True)
            continue
        if rec is None:
            rec = {"name": name, "X": X, "Ytot": Y, "points": []}
            doc["lattices"].append(rec)
        rec["points"] = []  # (the oracle's state is not kept between runs: a partial record starts over)
        if os.environ.get("ISING_GOLDEN_THREADS"):
            oracle.set_threads(int(os.environ["ISING_GOLDEN_THREADS"]))
        L = oracle.OracleLattice(X, Y, seed=1234, temp=oracle.CRIT_TEMP).init()
        t0 = time.time()
        for s in POINTS:
            L.sweep(s - L.it)
            up, dw = L.count()
            pt = {"sweeps": s, "up": up, "down": dw}
            if s in (0, 25, 144):
                pt["bond_equal"] = L.bond_equal()
            rec["points"].append(pt)
            json.dump(doc, open(OUT + ".tmp", "w"), indent=1)
            os.replace(OUT + ".tmp", OUT)
            print(f"{name} {Y}x{X} sweeps {s}: {pt} [{time.time() - t0:.0f} s]", flush=True)
        del L
    print("done", flush=True)


if __name__ == "__main__":
    main()
