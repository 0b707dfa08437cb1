"""the temperature-sweep driver of cuIsing (susceptibility, Binder cumulant, specific heat from exact
integer moments) against the oracle's committed config-5 series, and the binary checkpoint (resume == uninterrupted)."""
import csv
import json
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLI = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")


def run(args, cwd=None):
    r = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_tsweep_config5_against_oracle_series(gpu, tmp_path):
    """BASELINE config 5 in ONE process: 8192^2, seed 1234, T = 1.50 .. 3.00 step 0.05 (31 points), per point a fresh
    lattice and 4 measurements 4 sweeps apart = the oracle's golden points at sweeps 4/8/12/16
    (tests/golden/tsweep_8192.json).  Every (up, down, bond) triple must match; the derived averages are recomputed here
    from the golden integers in exact rational arithmetic."""
    fx = json.load(open(os.path.join(GOLD, "tsweep_8192.json")))
    out = run(["-x", fx["X"], "-y", fx["Y"], "-s", fx["seed"], "--tsweep", "1.5,3.0,0.05,0,4,4", "--tsweep-out", "ts"], cwd=tmp_path)
    assert "Temperature sweep: 31 points" in out
    series = list(csv.DictReader(open(tmp_path / "ts.series.csv")))
    points = list(csv.DictReader(open(tmp_path / "ts.csv")))
    assert len(points) == 31 and len(series) == 31 * 4
    N = fx["X"] * fx["Y"]
    for k, ser in enumerate(fx["series"]):
        rows = series[4 * k:4 * k + 4]
        gold = [p for p in ser["points"] if p["sweeps"] in (4, 8, 12, 16)]
        for row, g in zip(rows, gold):
            assert int(row["temp_bits"]) == ser["temp_bits"], (k, row)
            assert (int(row["iter"]), int(row["up"]), int(row["down"]), int(row["bond_equal"])) == (g["sweeps"], g["up"], g["down"], g["bond_equal"])
        Ms = [g["up"] - g["down"] for g in gold]
        Es = [2 * N - 2 * g["bond_equal"] for g in gold]
        pt = points[k]
        assert int(pt["temp_bits"]) == ser["temp_bits"] and int(pt["nmeas"]) == 4
        assert int(pt["sum_M"]) == sum(Ms) and int(pt["sum_absM"]) == sum(abs(m) for m in Ms)
        assert int(pt["sum_M2"]) == sum(m * m for m in Ms)
        assert int(pt["sum_E"]) == sum(Es) and int(pt["sum_E2"]) == sum(e * e for e in Es)
        T = Fraction(float(np.uint32(ser["temp_bits"]).view(np.float32)))
        n = 4
        mabs, m2 = Fraction(sum(abs(m) for m in Ms), n * N), Fraction(sum(m * m for m in Ms), n * N * N)
        m4 = Fraction(sum(m ** 4 for m in Ms), n * N ** 4)
        e1, e2 = Fraction(sum(Es), n * N), Fraction(sum(e * e for e in Es), n * N * N)
        want = {"m_abs": mabs, "m2": m2, "chi": N * (m2 - mabs * mabs) / T, "U4": 1 - m4 / (3 * m2 * m2), "e": e1,
                "Cv": N * (e2 - e1 * e1) / (T * T)}
        for key, val in want.items():
            assert float(pt[key]) == pytest.approx(float(val), rel=1e-9, abs=1e-12), (k, key)  # FP tolerance of the final ratios
        assert f"T = {float(T):f}: <|m|> = {float(mabs):9.6f}," in out
    # lower temperature -> lower energy after the same number of sweeps
    es = [float(p["e"]) for p in points]
    assert es[0] < es[-1] < 0


@pytest.mark.parametrize("extra", [["--tsweep-replicas", "1"], ["--tsweep-replicas", "4"], ["-J", "0.25"],
                                   ["--layout", "ballot"], ["--layout", "ballot", "--tsweep-replicas", "4"], ["--layout", "ballot", "--tsweep-no-batch"]])
def test_tsweep_replicas_give_every_point_the_series_of_a_run_of_its_own(gpu, tmp_path, extra):
    """Fresh-start mode simulates several temperature points together -- batched launches on the ballot layout (all ten
    points in one batch, or 4 + 4 + 2), otherwise side by side on streams of their own; the transcript and both CSV files
    must not depend on which, nor on how many at a time."""
    args = ["-x", 8192, "-y", 512, "-s", 99, "--tsweep", "1.8,2.7,0.1,33,3,5"]
    ref = run(args + ["--tsweep-out", "a"], cwd=tmp_path)
    if extra[0] == "-J":
        args += extra
        ref = run(args + ["--tsweep-replicas", "1", "--tsweep-out", "a"], cwd=tmp_path)
        got = run(args + ["--tsweep-replicas", "3", "--tsweep-out", "b"], cwd=tmp_path)
    else:
        got = run(args + extra + ["--tsweep-out", "b"], cwd=tmp_path)
    pick = lambda out: [ln for ln in out.splitlines() if ln.startswith("T = ")]
    assert len(pick(ref)) == 10 and pick(got) == pick(ref)
    for ext in (".csv", ".series.csv"):
        assert open(tmp_path / ("a" + ext)).read() == open(tmp_path / ("b" + ext)).read()


@pytest.mark.parametrize("extra", [[], ["--layout", "ballot"], ["--layout", "ballot", "--tsweep-no-batch"], ["--layout", "ballot", "--tsweep-replicas", "2"]])
def test_tsweep_chains_are_runs_of_their_own_with_error_bars(gpu, tmp_path, extra):
    """--tsweep-chains K simulates every temperature K times from seeds s .. s + K - 1 (all lattices in the same batched launches
    where the layout allows): chain c's rows of both CSV files are those of a single-chain run with seed s + c, and the summary
    file holds the mean of the chains' averages with their standard error."""
    import math
    base = ["-x", 8192, "-y", 512, "--tsweep", "2.4,2.6,0.1,33,4,5"] + extra
    K, seed = 3, 70
    run(base + ["-s", seed, "--tsweep-chains", K, "--tsweep-out", "all"], cwd=tmp_path)
    rows = [ln.split(",") for ln in open(tmp_path / "all.csv").read().splitlines()]
    head, rows = rows[0], rows[1:]
    assert head[-2:] == ["chain", "seed"] and len(rows) == 3 * K
    series = [ln.split(",") for ln in open(tmp_path / "all.series.csv").read().splitlines()[1:]]
    for c in range(K):
        run(base + ["-s", seed + c, "--tsweep-out", f"one{c}"], cwd=tmp_path)
        one = [ln.split(",") for ln in open(tmp_path / f"one{c}.csv").read().splitlines()[1:]]
        mine = [r[:-2] for r in rows if r[-2:] == [str(c), str(seed + c)]]
        assert mine == one, c
        one_series = [ln.split(",") for ln in open(tmp_path / f"one{c}.series.csv").read().splitlines()[1:]]
        assert [r[:-1] for r in series if r[-1] == str(c)] == one_series, c
    summ = [ln.split(",") for ln in open(tmp_path / "all.chains.csv").read().splitlines()]
    shead, summ = summ[0], summ[1:]
    assert len(summ) == 3
    for k, srow in enumerate(summ):
        assert int(srow[shead.index("chains")]) == K
        for name in ("m_abs", "e", "chi", "Cv", "U4", "m2"):
            vals = [float(r[head.index(name)]) for r in rows[k * K:(k + 1) * K]]
            mean = sum(vals) / K
            err = math.sqrt(sum((v - mean) ** 2 for v in vals) / (K - 1) / K)
            assert abs(float(srow[shead.index(name)]) - mean) <= 1e-12 * max(1.0, abs(mean)), name
            assert abs(float(srow[shead.index(name + "_err")]) - err) <= 1e-9 * abs(mean) + 1e-6 * err, name


@pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE])
def test_enqueued_measurements_equal_count_and_bond_sum(gpu, layout):
    """ising_measure_enqueue / _fetch (what --tsweep reads its series with): same integers as the blocking calls, for
    slabs side by side on private streams."""
    slabs = [ig.IsingSlab(8192, 96, seed=5 + k, temp=2.0 + 0.2 * k, layout=layout).use_private_stream().init() for k in range(3)]
    try:
        want = [[] for _ in slabs]
        for m in range(5):
            for k, s in enumerate(slabs):
                s.sweep(3)
                s.measure_enqueue()
        got = [s.measure_fetch() for s in slabs]
        assert all(s.measure_fetch() == [] for s in slabs)
        for k, s in enumerate(slabs):  # the same runs again, read with the blocking calls
            with ig.IsingSlab(8192, 96, seed=5 + k, temp=2.0 + 0.2 * k, layout=layout) as r:
                r.init()
                for m in range(5):
                    r.sweep(3)
                    want[k].append(r.count() + (r.bond_equal(),))
        assert got == want
    finally:
        for s in slabs:
            s.close()


def test_tsweep_anneal_matches_set_temperature_sequence(gpu, tmp_path):
    X, Y, seed = 4096, 256, 77
    run(["-x", X, "-y", Y, "-s", seed, "--tsweep", "2.0,2.5,0.25,3,2,2", "--tsweep-anneal", "--tsweep-out", "an"], cwd=tmp_path)
    series = list(csv.DictReader(open(tmp_path / "an.series.csv")))
    with ig.IsingSlab(X, Y, seed=seed, temp=2.0) as s:
        s.init()
        got = []
        for t in (2.0, 2.25, 2.5):
            s.set_temperature(t)
            s.sweep(3)
            for _ in range(2):
                s.sweep(2)
                got.append((s.it,) + s.count() + (s.bond_equal(),))
    assert [(int(r["iter"]), int(r["up"]), int(r["down"]), int(r["bond_equal"])) for r in series] == got


@pytest.mark.parametrize("layout", ["ballot", "dense", "nibble"])
def test_cli_checkpoint_resume_equals_uninterrupted(gpu, tmp_path, layout):
    X, Y, seed = 8192, 128, 4242
    base = ["-x", X, "-y", Y, "-s", seed, "-a", "1", "--layout", layout, "--energy"]
    full = run(base + ["-n", 16, "-p", 4, "-o"], cwd=tmp_path)
    (tmp_path / "part").mkdir()
    first = run(base + ["-n", 10, "-p", 4, "--checkpoint", "state.ckpt"], cwd=tmp_path / "part")
    assert "Checkpoint written to state.ckpt (10 iterations done)" in first
    second = run(["--resume", "state.ckpt", "-n", 6, "-p", 4, "-o", "--layout", layout, "--energy"], cwd=tmp_path / "part")
    assert "Resumed from state.ckpt: 10 iterations done" in second
    def lines(out):  # the magnetisation and energy lines of iterations 12 and 16 (without the "Final" repetition)
        return [ln for ln in out.splitlines() if ln.startswith("        ") and ("(iter:       12)" in ln or "(iter:       16)" in ln)]
    assert len(lines(full)) == 4 and lines(full) == lines(second)
    final = [ln for ln in full.splitlines() if ln.startswith("Final")]
    assert final and final == [ln for ln in second.splitlines() if ln.startswith("Final")]
    dump = f"lattice_{Y}x{X}_T_{ig.CRIT_TEMP_F32:f}_IT_{16:08d}_0.txt"
    assert (tmp_path / dump).read_bytes() == (tmp_path / "part" / dump).read_bytes()


def test_checkpoint_is_decomposition_independent(gpu, tmp_path):
    """Written by 4 slabs, loaded into 1 and into 2 slabs: identical continuation."""
    X, Ytot, seed, temp = 8192, 256, 9, ig.CRIT_TEMP_F32
    ring4 = ig.SlabSet([ig.IsingSlab(X, Ytot // 4, seed=seed, temp=temp, nslabs=4, slab=k) for k in range(4)]).init()
    ring4.sweep(5)
    path = tmp_path / "four.ckpt"
    ring4.checkpoint_save(path)
    ring4.sweep(3)
    want = (ring4.count(), ring4.bond_equal())
    ring4.close()
    assert ig.checkpoint_info(path)["it"] == 5 and ig.checkpoint_info(path)["Y_total"] == Ytot
    for n in (1, 2):
        # created at ANOTHER temperature: the load continues at the checkpoint's (a resume must not silently change T)
        ring = ig.SlabSet([ig.IsingSlab(X, Ytot // n, seed=seed, temp=(1.5 if n == 2 else temp), nslabs=n, slab=k) for k in range(n)])
        ring.checkpoint_load(path)
        assert ring.it == 5
        ring.sweep(3)
        assert (ring.count(), ring.bond_equal()) == want
        ring.close()
    with ig.IsingSlab(X, Ytot, seed=seed + 1, temp=temp) as s:  # another seed: the Philox streams would not continue
        with pytest.raises(ig.IsingError, match="seed"):
            ig.SlabSet([s]).checkpoint_load(path)
    bad = tmp_path / "bad.ckpt"
    data = bytearray(path.read_bytes())
    data[4096] ^= 0x10
    bad.write_bytes(bytes(data))
    with ig.IsingSlab(X, Ytot, seed=seed, temp=temp) as s:
        with pytest.raises(ig.IsingError, match="damaged"):
            ig.SlabSet([s]).checkpoint_load(bad)


@pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE])
def test_bits_and_packed_boundary_formats_round_trip(gpu, oracle_mod, layout):
    X, Y, seed = 8192, 96, 3
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=2.0).init().sweep(2)

    def bits_of(packed):  # oracle packed rows -> the 1 bit/spin boundary format
        nib = ((packed[..., None] >> (4 * np.arange(16, dtype=np.uint64))) & np.uint64(1)).astype(np.uint32)  # [Y][lld][16]
        half = (nib << np.arange(16, dtype=np.uint32)).sum(axis=-1, dtype=np.uint32)                            # 16 bits per word
        return half[:, 0::2] | (half[:, 1::2] << np.uint32(16))

    with ig.IsingSlab(X, Y, seed=seed, temp=2.0, layout=layout) as s:
        s.init().sweep(2)
        for color, ref in ((ig.BLACK, orc.black), (ig.WHITE, orc.white)):
            assert np.array_equal(s.read(color), ref)
            assert np.array_equal(s.read_bits(color), bits_of(ref))
            assert np.array_equal(s.read(color, 17, 40), ref[17:57])
    # write the oracle's state into a fresh slab through either format, continue, compare
    orc.sweep(3)
    for writer in ("packed", "bits"):
        ref = oracle_mod.OracleLattice(X, Y, seed=seed, temp=2.0).init().sweep(2)
        with ig.IsingSlab(X, Y, seed=seed, temp=2.0, layout=layout) as s:
            for color, rows in ((ig.BLACK, ref.black), (ig.WHITE, ref.white)):
                if writer == "packed":
                    s.write(color, rows[:50]); s.write(color, rows[50:], 50)
                else:
                    s.write_bits(color, bits_of(rows))
            s.it = 2
            s.sweep(3)
            assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)


def test_tsweep_small_lattices_ride_in_batched_passes(gpu, tmp_path):
    """Round 6: a temperature series on lattices of the quad path (2048^2: 64 tiles for 256 CUs each) runs as ONE batch -- the tiles of all 31 x 2 lattices in one
    launch per pass, the measurements inside the passes (ising_batch_sweep_counted) --, and every (up, down, bond) triple is the oracle's for that temperature
    and seed (tests/golden/tsweep_2048.json: seeds 1234 / 1235 = --tsweep-chains 2, after 8, 16, 24, 32 sweeps)."""
    fx = json.load(open(os.path.join(GOLD, "tsweep_2048.json")))
    r = subprocess.run([CLI, "-x", str(fx["X"]), "-y", str(fx["Y"]), "-s", str(fx["seeds"][0]), "--tsweep", "1.5,3.0,0.05,0,4,8", "--tsweep-chains", "2", "--tsweep-out", "ts"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if "per batched launch (tiles of" not in r.stderr:
        pytest.skip("the quad path is the default on a whole MI355X only: " + r.stderr[-300:])
    series = list(csv.DictReader(open(tmp_path / "ts.series.csv")))
    assert len(series) == 31 * 2 * 4
    gold = {(s["seed"], s["temp_bits"], p["sweeps"]): p for s in fx["series"] for p in s["points"]}
    for row in series:
        g = gold[(fx["seeds"][0] + int(row["chain"]), int(row["temp_bits"]), int(row["iter"]))]
        assert (int(row["up"]), int(row["down"]), int(row["bond_equal"])) == (g["up"], g["down"], g["bond_equal"]), row
    assert {int(row["iter"]) for row in series} == {8, 16, 24, 32}


@pytest.mark.parametrize("extra", [[], ["--tsweep-replicas", "7"], ["--tsweep-no-batch"]])
def test_tsweep_4096_against_oracle_series(gpu, tmp_path, extra):
    """... 4096^2 (two blocks of 2048 columns), eight sweeps of equilibration and one measurement eight sweeps later = the golden's point at sixteen sweeps; groups of
    seven points (a last group of three) and the unbatched form give the same integers."""
    fx = json.load(open(os.path.join(GOLD, "tsweep_4096.json")))
    run(["-x", fx["X"], "-y", fx["Y"], "-s", fx["seeds"][0], "--tsweep", "1.5,3.0,0.05,8,1,8", "--tsweep-out", "ts"] + extra, cwd=tmp_path)
    series = list(csv.DictReader(open(tmp_path / "ts.series.csv")))
    assert len(series) == 31
    for row, s in zip(series, fx["series"]):
        g = [p for p in s["points"] if p["sweeps"] == 16][0]
        assert int(row["temp_bits"]) == s["temp_bits"]
        assert (int(row["iter"]), int(row["up"]), int(row["down"]), int(row["bond_equal"])) == (16, g["up"], g["down"], g["bond_equal"])
