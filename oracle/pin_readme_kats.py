"""Pin the CPU oracle against EVERY golden vector the reference holds for the hot path.

The reference has no tests; its only result-pinning artefacts are the console transcripts in
/root/reference/optimized/README.md.  This script replays them with the CPU oracle and records what the
oracle produced next to what the README shows, in tests/golden/readme_kat_pin.json (committed).  It is slow
(tens of minutes on 8 cores) so it is run by hand in the dev container; tests/test_oracle_kat.py re-checks
the cheap subset on every CPU test run and verifies the committed pin file says "all matched".

Usage: python oracle/pin_readme_kats.py [--quick] [--cases 2,3]
  --cases: replay only these cases (0-based) and merge them into the committed pin file (the 8-GPU lattice, 2^35 spins
  = 16 GiB packed, takes about an hour per 128 sweeps on 8 cores; the file is rewritten after every check point).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402

# (name, X, Ytot, XSL, YSL, temp, {iter: (up, dw)}) -- numbers typed from optimized/README.md (file:line in name)
CASES = [
    ("README.md:94-137  2xV100 -y 32768 -x 65536 -d 2 -t 1.5", 65536, 65536, 0, 0, 1.5, {
        0: (2147484090, 2147483206), 16: (2147575418, 2147391878), 32: (2147641872, 2147325424),
        48: (2147605659, 2147361637), 64: (2147701147, 2147266149), 80: (2147558546, 2147408750),
        96: (2147471275, 2147496021), 112: (2147612509, 2147354787), 128: (2147678887, 2147288409)}),
    ("README.md:148-196 2xV100 same + --xsl 2048 --ysl 2048", 65536, 65536, 2048, 2048, 1.5, {
        0: (2147484090, 2147483206), 16: (2147594634, 2147372662), 32: (2147631783, 2147335513),
        48: (2147550893, 2147416403), 64: (2147630364, 2147336932), 80: (2147500244, 2147467052),
        96: (2147357073, 2147610223), 112: (2147482936, 2147484360), 128: (2147461873, 2147505423)}),
    ("README.md:206-249 2xA100 -y 65536 -x 65536 -d 2 -t 1.5 (also :327-370 2xH100)", 65536, 131072, 0, 0, 1.5, {
        0: (4294989182, 4294945410), 16: (4294617248, 4295317344), 32: (4293898346, 4296036246),
        48: (4292806461, 4297128131), 64: (4291852263, 4298082329), 80: (4291086016, 4298848576),
        96: (4290256223, 4299678369), 112: (4289621029, 4300313563), 128: (4288877118, 4301057474)}),
    ("README.md:255-316 8xA100 -y 65536 -x 65536 -d 8 -t 1.5 (also :375-436 8xH100)", 65536, 524288, 0, 0, 1.5, {
        0: (17179689306, 17180049062), 16: (17176389528, 17183348840), 32: (17172963073, 17186775295),
        48: (17170610910, 17189127458), 64: (17168843228, 17190895140), 80: (17167009008, 17192729360),
        96: (17165014291, 17194724077), 112: (17163708078, 17196030290), 128: (17162287230, 17197451138)}),
]


def main():
    quick = "--quick" in sys.argv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "readme_kat_pin.json")
    only = None
    if "--cases" in sys.argv:
        only = [int(v) for v in sys.argv[sys.argv.index("--cases") + 1].split(",")]
    out = {"generated_by": "oracle/pin_readme_kats.py", "cases": [], "all_matched": True}
    if only is not None:  # merge into the committed file
        out = json.load(open(path))
        assert len(out["cases"]) == len(CASES)

    def write():
        out["all_matched"] = all(c["match"] for r in out["cases"] for c in r["checks"])
        with open(path + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(path + ".tmp", path)

    for idx, (name, X, Y, XSL, YSL, temp, want) in enumerate(CASES):
        if only is not None and idx not in only:
            continue
        t0 = time.time()
        L = oracle.OracleLattice(X, Y, seed=oracle.SEED_DEF, temp=temp, XSL=XSL, YSL=YSL).init()
        rec = {"name": name, "X": X, "Ytot": Y, "XSL": XSL, "YSL": YSL, "temp": temp, "seed": oracle.SEED_DEF, "checks": []}
        last = max(want) if not quick else min(16, max(want))
        for it in sorted(want):
            if it > last:
                break
            if it > L.it:
                L.sweep(it - L.it)
            got = L.count()
            ok = tuple(got) == tuple(want[it])
            rec["checks"].append({"iter": it, "readme": list(want[it]), "oracle": list(got), "match": ok})
            out["all_matched"] &= ok
            print(f"{name[:40]:40s} it {it:4d} oracle {got} readme {want[it]} {'OK' if ok else 'MISMATCH'} [{time.time()-t0:.0f}s]", flush=True)
            if only is not None and not quick:  # long runs: keep what has been checked so far
                rec["seconds"] = round(time.time() - t0, 1)
                out["cases"][idx] = rec
                write()
        rec["seconds"] = round(time.time() - t0, 1)
        if only is not None:
            out["cases"][idx] = rec
        else:
            out["cases"].append(rec)
        del L
    if not quick:
        write()
        print("wrote", os.path.normpath(path))
    print("ALL MATCHED" if out["all_matched"] else "MISMATCHES PRESENT")


if __name__ == "__main__":
    main()
