#!/usr/bin/env python3
"""BASELINE config 5's temperature series against the exact solution of the infinite square lattice: `cuIsing --tsweep` from the
ordered start (--tsweep-cold) with K independently seeded chains per temperature (--tsweep-chains K: all 31 x K lattices advance in
batched launches), <|m|> next to Yang's spontaneous magnetisation and <e> next to Onsager's energy, deviations in units of the
chains' standard error.  Near T_c (2.2 .. 2.45) the relaxation time and the finite lattice show; elsewhere the curve must sit
on the exact one.   Usage: python tools/curve_vs_exact.py [L=8192] [K=4] [nequil=1000] [nmeas=20] [stride=10]"""
import math
import os
import subprocess
import sys
import tempfile

from scipy.special import ellipk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ising_gpu_amd", "cuIsing")
L, K, nequil, nmeas, stride = (int(v) for v in (sys.argv[1:6] + ["8192", "4", "1000", "20", "10"][len(sys.argv) - 1:]))


def onsager_energy(T):
    b = 1.0 / T
    k = 2.0 * math.sinh(2 * b) / math.cosh(2 * b) ** 2
    return -(1.0 / math.tanh(2 * b)) * (1.0 + (2.0 / math.pi) * (2.0 * math.tanh(2 * b) ** 2 - 1.0) * ellipk(k * k))


def yang_magnetisation(T):
    s = math.sinh(2.0 / T) ** -4
    return (1.0 - s) ** 0.125 if s < 1.0 else 0.0


with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run([CLI, "-x", str(L), "-y", str(L), "-s", "20260929", "--tsweep", f"1.5,3.0,0.05,{nequil},{nmeas},{stride}", "--tsweep-cold",
                        "--tsweep-chains", str(K), "--tsweep-out", "curve"], capture_output=True, text=True, cwd=tmp, timeout=1800)
    if r.returncode:
        raise SystemExit(r.stderr)
    rows = [ln.split(",") for ln in open(os.path.join(tmp, "curve.chains.csv")).read().splitlines()]
head, rows = rows[0], rows[1:]
col = {n: head.index(n) for n in head}
print(f"# {L} x {L}, {K} chains per temperature from the ordered lattice, {nequil} equilibration + {nmeas} x {stride} measurement sweeps")
print([ln for ln in r.stdout.splitlines() if "flips/ns" in ln][-1].strip())
print("# (the K chains use the same K seeds at every temperature: deviations are correlated across T -- common random numbers)")
print(f"{'T':>6} {'<|m|>':>10} {'+-':>9} {'Yang':>10} {'dev/err':>8}   {'<e>':>10} {'+-':>9} {'Onsager':>10} {'dev/err':>8}")
for row in rows:
    T = float(row[col["temp"]])
    m, dm, e, de = (float(row[col[n]]) for n in ("m_abs", "m_abs_err", "e", "e_err"))
    ym, oe = yang_magnetisation(T), onsager_energy(T)
    mark = "   (critical region)" if 2.175 < T < 2.475 else ""
    mdev = f"{(m - ym) / max(dm, 1e-12):8.1f}" if ym > 0 else f"{'-':>8}"  # (above T_c <|m|> of a finite lattice is ~ sqrt(chi T / N), not 0)
    print(f"{T:6.3f} {m:10.6f} {dm:9.6f} {ym:10.6f} {mdev}   {e:10.6f} {de:9.6f} {oe:10.6f} {(e - oe) / max(de, 1e-12):8.1f}{mark}")
