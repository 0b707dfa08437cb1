"""BASELINE config 2 at FULL length from the pinned CPU oracle: 16384 x 16384, T = CRIT_TEMP, seed 1234, 10^5 sweeps --
counts, bond sum and SHA-256 of the packed state every 10000 sweeps (and at 4096, which config2_16384.json already holds:
the run checks itself against it on the way).  2.7e13 spin updates: ~83 min on the 16 host cores a GPU box grants
(5.4 flips/ns; `gpurun -- python tests/golden/make_golden_config2_full.py --out gpurun_out/config2_16384_full.json --gpu`),
~10 h on the dev container's 8.  The file is rewritten after every point, so a run that is cut short leaves a valid prefix.

  --gpu   also sweep the lattice on the GPU (ising_gpu_amd, default layout) and compare at every point -- a report on
          stdout only; the golden file holds oracle numbers and nothing else.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(HERE, "config2_16384_full.json"))
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--points", default="4096," + ",".join(str(k) for k in range(10000, 100001, 10000)))
ap.add_argument("--gpu", action="store_true")
args = ap.parse_args()

X = Y = 16384
SEED = 1234
oracle.set_threads(args.threads)
temp = np.float32(oracle.CRIT_TEMP)
out = {"X": X, "Ytot": Y, "seed": SEED, "temp": float(temp), "temp_bits": int(temp.view(np.uint32)), "points": []}
prefix = {p["sweeps"]: p for p in json.load(open(os.path.join(HERE, "config2_16384.json")))["points"]}


def sha(black, white):
    h = hashlib.sha256()
    h.update(black.tobytes())
    h.update(white.tobytes())
    return h.hexdigest()


L = oracle.OracleLattice(X, Y, seed=SEED, temp=oracle.CRIT_TEMP).init()
gpu = None
if args.gpu:
    import ising_gpu_amd as ig
    gpu = ig.IsingSlab(X, Y, seed=SEED, temp=ig.CRIT_TEMP_F32).init()
t0 = time.time()
for s in (int(v) for v in args.points.split(",")):
    L.sweep(s - L.it)
    up, dw = L.count()
    rec = {"sweeps": s, "up": up, "down": dw, "bond_equal": L.bond_equal(), "sha256": sha(L.black, L.white)}
    if s in prefix:
        assert rec == prefix[s], (rec, prefix[s])
    out["points"].append(rec)
    json.dump(out, open(args.out, "w"), indent=1)
    line = f"sweeps {s}: {rec}  [{time.time() - t0:.0f} s]"
    if gpu is not None:
        gpu.sweep(s - gpu.it)
        same = (gpu.count() == (up, dw) and gpu.bond_equal() == rec["bond_equal"] and
                sha(gpu.read(ig.BLACK), gpu.read(ig.WHITE)) == rec["sha256"])
        line += f"  GPU {'==' if same else '!='} oracle (counts, bond sum, SHA-256 of the packed state)"
    print(line, flush=True)
print("done")
