#!/usr/bin/env python3
"""A/B of library builds and environment settings on lone lattices: flips/ns of ising_sweep, best of 3 after a preheat, every cell checked
against the first cell of its lattice (counts + bond sum after 96 sweeps from the seed).

usage: ab_probe.py [--libs name=path,...] [--env "K=V K=V;K=V ..."] [--shapes XxY,...] [--H 0,4,8] [--wgs 0,4,5,6]
  --libs    builds to compare (default: the product library); a name without a path = ising_gpu_amd/libising_hip_<name>.so
  --env     semicolon-separated environment variants (each a space-separated list of K=V), applied on top of every lib
  --H       strip heights (ising_config.strip_rows; 0 = the library's choice)
  --wgs     workgroups per CU of fused launches (ISING_FUSED_WGS = 256 x; 0 = the library's choice)
One line per (lattice, H): cells by wgs, inside a cell lib x env in the order given, separated by '/'.
"""
import argparse
import os
import subprocess
import sys
import time
os.environ.setdefault("ISING_GUARD", "0")  # (a probe measures the shapes it asks for: the run-time guard would move them)

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "case":
    import ising_gpu_amd as ig
    X, Y, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sweeps = max(64, min(4096, (1 << 37) // (X * Y) // 32 * 32))
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, strip_rows=H) as s:
        s.init()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            s.sweep(64)
            s.synchronize()
        s.init().sweep(96)
        chk = (s.count(), s.bond_equal())
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            s.sweep(sweeps)
            s.synchronize()
            best = max(best, X * Y * sweeps / (time.perf_counter() - t0) * 1e-9)
        print("RESULT", best, s.strip_rows, s.layout, chk[0][0], chk[1])
    sys.exit(0)

ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="product")
ap.add_argument("--env", default="")
ap.add_argument("--shapes", default="8192x8192,16384x8192,16384x16384,65536x8192")
ap.add_argument("--H", default="0")
ap.add_argument("--wgs", default="0")
args = ap.parse_args()
libs = []
for item in args.libs.split(","):
    name, _, path = item.partition("=")
    if not path:
        path = os.path.join(ROOT, "ising_gpu_amd", "libising_hip.so" if name == "product" else f"libising_hip_{name}.so")
    libs.append((name, path))
envs = [dict(kv.split("=", 1) for kv in v.split()) for v in args.env.split(";")] if args.env else [{}]
print("cells: " + " / ".join(f"{n}{' ' + ' '.join(f'{k}={v}' for k, v in e.items()) if e else ''}" for n, _ in libs for e in envs), flush=True)
for shape in args.shapes.split(","):
    X, Y = map(int, shape.split("x"))
    print(f"{Y} rows x {X} columns: rows = strip height (0 = the library's choice), columns = workgroups per CU {args.wgs} (0 = the library's choice)", flush=True)
    ref = None
    for H in map(int, args.H.split(",")):
        if H and Y % H:
            continue
        row = []
        for per_cu in map(int, args.wgs.split(",")):
            cell = []
            for name, path in libs:
                for e in envs:
                    env = dict(os.environ, ISING_LIB=path, ISING_ABORT_POLLS="40000", **e)
                    if per_cu:
                        env["ISING_FUSED_WGS"] = str(256 * per_cu)
                    try:
                        r = subprocess.run([sys.executable, __file__, "case", str(X), str(Y), str(H)], env=env, capture_output=True, text=True, timeout=600)
                        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
                    except subprocess.TimeoutExpired:
                        res = []
                    if not res:
                        cell.append(" FAILED")
                        continue
                    f = res[-1].split()
                    ref = ref or f[4:]
                    cell.append(f"{float(f[1]):7.1f}" + ("" if f[4:] == ref else "!") + (f"(H{f[2]})" if H == 0 and per_cu == 0 and len(cell) == 0 else ""))
            row.append("/".join(cell))
        print(f"  H = {H:2d}: " + "   ".join(row), flush=True)
