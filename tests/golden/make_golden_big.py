"""Generate the full-size golden fixtures from the pinned CPU oracle (run in the dev container: ~45 min on 8 cores,
up to 17 GiB of host memory; outputs are numbers only).

  bench_65536_tc.json  -- the bench.py workload (BASELINE config 3): 65536 x 65536, T = CRIT_TEMP, seed 1234:
                          (up, down, bond-equal) after 0, 1, 2, 5, 16, 21, 25, 32, 64, 128, 144 sweeps
  ring_65536_tc.json   -- bench.py --gpus N: (N*65536) x 65536 rows x columns, same seed/temperature, N = 2, 4, 8:
                          counts after 0, 5, 21, 25 sweeps
  config4_131072.json  -- BASELINE config 4: 131072 x 131072 (8 slabs of 16384 rows), T = CRIT_TEMP, seed 1234:
                          counts, bond-equal and the up-count of each of the 8 slabs after 0, 1, 2 sweeps
  config2_16384.json   -- BASELINE config 2: 16384 x 16384, T = CRIT_TEMP, seed 1234: counts / bond / SHA-256 of the
                          packed state after 0, 1, 2, 4, 16, 64, 256, 1024, 4096 sweeps (prefix of the 10^5-sweep run)

Usage: make_golden_big.py [bench] [ring] [config4] [config2]   (default: all)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def _series(X, Ytot, seed, temp, points, bond=True, extra=None):
    L = oracle.OracleLattice(X, Ytot, seed=seed, temp=temp).init()
    out = []
    t0 = time.time()
    for s in points:
        L.sweep(s - L.it)
        up, dw = L.count()
        rec = {"sweeps": s, "up": up, "down": dw}
        if bond:
            rec["bond_equal"] = L.bond_equal()
        if extra:
            rec.update(extra(L))
        out.append(rec)
        print(f"  {Ytot}x{X} sweeps {s}: {rec}  [{time.time() - t0:.0f} s]", flush=True)
    return out


def _hdr(X, Ytot, seed=1234):
    t = np.float32(oracle.CRIT_TEMP)
    return {"X": X, "Ytot": Ytot, "seed": seed, "temp": float(t), "temp_bits": int(t.view(np.uint32))}


def bench():
    out = _hdr(65536, 65536)
    out["points"] = _series(65536, 65536, 1234, oracle.CRIT_TEMP, (0, 1, 2, 5, 16, 21, 25, 32, 64, 128, 144))
    json.dump(out, open(os.path.join(HERE, "bench_65536_tc.json"), "w"), indent=1)


def ring():
    out = {"rows_per_slab": 65536, "rings": []}
    for n in (2, 4, 8):
        rec = _hdr(65536, 65536 * n)
        rec["nslabs"] = n
        rec["points"] = _series(65536, 65536 * n, 1234, oracle.CRIT_TEMP, (0, 5, 21, 25), bond=False)
        out["rings"].append(rec)
        json.dump(out, open(os.path.join(HERE, "ring_65536_tc.json"), "w"), indent=1)


def config4():
    def per_slab(L):
        Y = L.Y // 8
        ups = []
        for k in range(8):
            ups.append(int(sum(int(np.unpackbits(a[k * Y:(k + 1) * Y].view(np.uint8)).sum(dtype=np.int64)) for a in (L.black, L.white))))
        return {"slab_up": ups}
    out = _hdr(131072, 131072)
    out["nslabs"] = 8
    out["points"] = _series(131072, 131072, 1234, oracle.CRIT_TEMP, (0, 1, 2), extra=per_slab)
    json.dump(out, open(os.path.join(HERE, "config4_131072.json"), "w"), indent=1)


def config2():
    def sha(L):
        h = hashlib.sha256()
        h.update(L.black.tobytes())
        h.update(L.white.tobytes())
        return {"sha256": h.hexdigest()}
    out = _hdr(16384, 16384)
    out["points"] = _series(16384, 16384, 1234, oracle.CRIT_TEMP, (0, 1, 2, 4, 16, 64, 256, 1024, 4096), extra=sha)  # (4096: ~15 min on 8 cores)
    json.dump(out, open(os.path.join(HERE, "config2_16384.json"), "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["config2", "config4", "bench", "ring"]
    for w in what:
        print("==", w, flush=True)
        {"bench": bench, "ring": ring, "config4": config4, "config2": config2}[w]()
    print("done")
