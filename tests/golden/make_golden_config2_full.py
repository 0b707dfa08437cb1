"""BASELINE config 2 at FULL length from the pinned CPU oracle: 16384 x 16384, T = CRIT_TEMP, seed 1234, 10^5 sweeps --
counts, bond sum and SHA-256 of the packed state every 5000 sweeps (and at 4096, which config2_16384.json already holds:
the run checks itself against it on the way).  2.7e13 spin updates: ~83 min on the 16 host cores a GPU box grants
(5.4 flips/ns), ~10 h on the dev container's 8.  A gpurun call lasts an hour at most, so the run is resumable: after every
point the oracle's state goes to --state (1 bit per spin, 32 MiB) and the points so far to --out; a later call with the same
two files checks the state against the SHA-256 of the last point and goes on from there.

  gpurun --timeout 3600 -- 'timeout 3400 python tests/golden/make_golden_config2_full.py --out gpurun_out/config2_16384_full.json \
                            --state gpurun_out/config2_state.bits --gpu'      (twice; then copy the .json to tests/golden/)

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
ap.add_argument("--state", default=None, help="resume file: the oracle's state at the last point of --out, 1 bit per spin")
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--points", default="4096," + ",".join(str(k) for k in range(5000, 100001, 5000)))
ap.add_argument("--gpu", action="store_true")
args = ap.parse_args()

X = Y = 16384
SEED = 1234
oracle.set_threads(args.threads)
temp = np.float32(oracle.CRIT_TEMP)
hdr = {"X": X, "Ytot": Y, "seed": SEED, "temp": float(temp), "temp_bits": int(temp.view(np.uint32))}
prefix = {p["sweeps"]: p for p in json.load(open(os.path.join(HERE, "config2_16384.json")))["points"]}


def sha(black, white):
    h = hashlib.sha256()
    h.update(black.tobytes())
    h.update(white.tobytes())
    return h.hexdigest()


def to_bits(words):
    """(Y, X/32) uint64, one spin per nibble -> the nibbles' low bits, 1 bit per spin."""
    b = np.unpackbits(words.view(np.uint8), bitorder="little").reshape(-1, 8)[:, (0, 4)]
    return np.packbits(b.reshape(-1), bitorder="little")


def from_bits(bits, shape):
    b = np.unpackbits(bits, bitorder="little").reshape(-1, 2)
    return (b[:, 0] | (b[:, 1] << 4)).astype(np.uint8).view(np.uint64).reshape(shape)


L = oracle.OracleLattice(X, Y, seed=SEED, temp=oracle.CRIT_TEMP).init()
out = dict(hdr, points=[])
if args.state and os.path.exists(args.state) and os.path.exists(args.out):
    have = json.load(open(args.out))
    assert {k: have[k] for k in hdr} == hdr
    raw = np.fromfile(args.state, dtype=np.uint8)
    at = int(raw[:8].view(np.uint64)[0])  # (the state is written after the points: it may be one point behind them)
    half = (raw.size - 8) // 2
    L.black[:] = from_bits(raw[8:8 + half], L.black.shape)
    L.white[:] = from_bits(raw[8 + half:], L.white.shape)
    have["points"] = [p for p in have["points"] if p["sweeps"] <= at]
    last = have["points"][-1]
    assert last["sweeps"] == at and sha(L.black, L.white) == last["sha256"], "the state file is not the state of a point of --out"
    L.it = at
    out = have
    print(f"resumed at sweep {L.it} (state file matches the SHA-256 of that point)", flush=True)

gpu = None
if args.gpu:
    import ising_gpu_amd as ig
    gpu = ig.IsingSlab(X, Y, seed=SEED, temp=ig.CRIT_TEMP_F32).init()
    if L.it:
        gpu.sweep(L.it)
        print(f"GPU at sweep {L.it}: SHA-256 {'==' if sha(gpu.read(ig.BLACK), gpu.read(ig.WHITE)) == sha(L.black, L.white) else '!='} the resumed oracle state", flush=True)
t0 = time.time()
for s in (int(v) for v in args.points.split(",")):
    if s <= L.it:
        continue
    L.sweep(s - L.it)
    up, dw = L.count()
    rec = {"sweeps": s, "up": up, "down": dw, "bond_equal": L.bond_equal(), "sha256": sha(L.black, L.white)}
    if s in prefix:
        assert rec == prefix[s], (rec, prefix[s])
    out["points"].append(rec)
    json.dump(out, open(args.out + ".tmp", "w"), indent=1)
    os.replace(args.out + ".tmp", args.out)
    if args.state:
        np.concatenate([np.array([s], dtype=np.uint64).view(np.uint8), to_bits(L.black), to_bits(L.white)]).tofile(args.state + ".tmp")
        os.replace(args.state + ".tmp", args.state)
    line = f"sweeps {s}: {rec}  [{time.time() - t0:.0f} s]"
    if gpu is not None:
        gpu.sweep(s - gpu.it)
        same = (gpu.count() == (up, dw) and gpu.bond_equal() == rec["bond_equal"] and
                sha(gpu.read(ig.BLACK), gpu.read(ig.WHITE)) == rec["sha256"])
        line += f"  GPU {'==' if same else '!='} oracle (counts, bond sum, SHA-256 of the packed state)"
    print(line, flush=True)
print("done" if L.it >= 100000 else f"stopped at {L.it}")
