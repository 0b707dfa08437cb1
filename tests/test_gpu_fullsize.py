"""BASELINE.json configurations at FULL size against the committed oracle goldens (tests/golden/make_golden_big.py):
  config 2  16384 x 16384, T = Tc, seed 1234         -- SHA-256 of the packed state, counts, bond sums
  config 3  65536 x 65536, T = Tc, seed 1234         -- the bench.py workload: counts and bond sums (0 .. 144 sweeps)
  config 4  131072 x 131072 as 8 slabs of 16384 rows -- counts, bond sums, per-slab counts; through the C-ABI ring
            (SlabSet), the torch-side LocalRing and `cuIsing -d 8` (all slabs on device 0 of a 1-GPU box)
plus a live comparison of config 2 with the oracle run on this box's host cores."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLI = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")


def _gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _f32(bits):
    return float(np.uint32(bits).view(np.float32))


@pytest.mark.parametrize("layout", [ig.LAYOUT_AUTO, ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE])
def test_config2_16384_golden_states(gpu, layout):
    fx = _gold("config2_16384.json")
    with ig.IsingSlab(fx["X"], fx["Ytot"], seed=fx["seed"], temp=_f32(fx["temp_bits"]), layout=layout) as s:
        s.init()
        for pt in fx["points"]:
            s.sweep(pt["sweeps"] - s.it)
            assert s.count() == (pt["up"], pt["down"]), pt["sweeps"]
            assert s.bond_equal() == pt["bond_equal"], pt["sweeps"]
            if pt["sweeps"] in (0, 2, 256, 4096):
                h = hashlib.sha256()
                h.update(s.read(ig.BLACK).tobytes())
                h.update(s.read(ig.WHITE).tobytes())
                assert h.hexdigest() == pt["sha256"], pt["sweeps"]


def test_config2_full_length_against_the_oracle(gpu):
    """BASELINE config 2 end to end: all 10^5 sweeps of the 16384^2 lattice, counts and bond sums at every point the oracle
    recorded on the way (tests/golden/make_golden_config2_full.py: 2.7e13 spin updates on the CPU), SHA-256 of the packed state
    half way and at the end."""
    fx = _gold("config2_16384_full.json")
    assert fx["points"][-1]["sweeps"] == 100000 and len(fx["points"]) >= 20
    with ig.IsingSlab(fx["X"], fx["Ytot"], seed=fx["seed"], temp=_f32(fx["temp_bits"])) as s:
        assert s.fused
        s.init()
        for pt in fx["points"]:
            s.sweep(pt["sweeps"] - s.it)
            assert s.count() == (pt["up"], pt["down"]), pt["sweeps"]
            assert s.bond_equal() == pt["bond_equal"], pt["sweeps"]
            if pt["sweeps"] in (50000, 100000):
                h = hashlib.sha256()
                h.update(s.read(ig.BLACK).tobytes())
                h.update(s.read(ig.WHITE).tobytes())
                assert h.hexdigest() == pt["sha256"], pt["sweeps"]


def test_config2_16384_full_state_vs_live_oracle(gpu, oracle_mod):
    X = Y = 16384
    orc = oracle_mod.OracleLattice(X, Y, seed=1234, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        for upto in (0, 1, 3):
            s.sweep(upto - s.it)
            orc.sweep(upto - orc.it)
            assert np.array_equal(s.read(ig.BLACK), orc.black), upto
            assert np.array_equal(s.read(ig.WHITE), orc.white), upto
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


def test_config3_bench_workload_golden(gpu):
    """65536^2 at T = Tc, seed 1234 -- exactly what bench.py times; 25 = the driver's 5 + 20 sweeps, 144 = the default
    16 + 128."""
    fx = _gold("bench_65536_tc.json")
    with ig.IsingSlab(fx["X"], fx["Ytot"], seed=fx["seed"], temp=_f32(fx["temp_bits"])) as s:
        assert s.layout == ig.LAYOUT_BALLOT
        s.init()
        for pt in fx["points"]:
            s.sweep(pt["sweeps"] - s.it)
            assert s.count() == (pt["up"], pt["down"]), pt["sweeps"]
            if pt["sweeps"] in (0, 25, 144):
                assert s.bond_equal() == pt["bond_equal"], pt["sweeps"]


def _config4_slabs(fx, **kw):
    n = fx["nslabs"]
    return [ig.IsingSlab(fx["X"], fx["Ytot"] // n, seed=fx["seed"], temp=_f32(fx["temp_bits"]), nslabs=n, slab=k, **kw) for k in range(n)]


def test_config4_131072_eight_slabs_capi_ring(gpu):
    fx = _gold("config4_131072.json")
    ring = ig.SlabSet(_config4_slabs(fx))
    try:
        ring.init()
        for pt in fx["points"]:
            ring.sweep(pt["sweeps"] - ring.it)
            assert [s.count()[0] for s in ring.slabs] == pt["slab_up"], pt["sweeps"]
            assert ring.count() == (pt["up"], pt["down"]), pt["sweeps"]
            assert ring.bond_equal() == pt["bond_equal"], pt["sweeps"]
    finally:
        ring.close()


def test_config4_131072_eight_slabs_local_ring(gpu):
    fx = _gold("config4_131072.json")
    slabs = _config4_slabs(fx)
    try:
        ring = ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init()
        for pt in fx["points"]:
            ring.sweep(pt["sweeps"] - ring.it)
            assert ring.count() == (pt["up"], pt["down"]), pt["sweeps"]
            assert ring.bond_equal() == pt["bond_equal"], pt["sweeps"]
    finally:
        for s in slabs:
            s.close()


def test_config4_131072_cli_eight_devices_mapped_to_one(gpu):
    """`cuIsing -x 131072 -y 16384 -d 8 -a 1 -s 1234 -n 2 -p 1` (BASELINE config 4's command line) with every slab on
    device 0: the reference's transcript lines with the oracle's counts."""
    fx = _gold("config4_131072.json")
    r = subprocess.run([CLI, "-x", "131072", "-y", "16384", "-d", "8", "-a", "1", "-s", "1234", "-n", "2", "-p", "1",
                        "--devmap", "0,0,0,0,0,0,0,0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    n = fx["X"] * fx["Ytot"]
    p0, p1, p2 = fx["points"]
    assert f"\ttotal lattice size:        131072 x   131072\n" in r.stdout
    assert f"Initial magnetization: {abs(p0['up'] - p0['down']) / n:9.6f}, up_s: {p0['up']:12d}, dw_s: {p0['down']:12d}\n" in r.stdout
    for pt in (p1, p2):
        assert (f"        magnetization: {abs(pt['up'] - pt['down']) / n:9.6f}, up_s: {pt['up']:12d}, dw_s: {pt['down']:12d} "
                f"(iter: {pt['sweeps']:8d})\n") in r.stdout


_RING_OF_ONE = {
    "rccl, ghost rows 64 deep (default)": ("native", {}),
    "rccl, one halo row, event schedule": ("native", {"ISING_RING_GHOST": "1"}),
    "rccl, caller-owned buffer": ("native-torch", {}),
    "copies, ghost rows, two streams": ("copy", {"ISING_RING_INLINE": "0", "ISING_RING_STORE": "0"}),
    "copies, one halo row, two streams": ("copy", {"ISING_RING_GHOST": "1", "ISING_RING_INLINE": "0", "ISING_RING_STORE": "0"}),
    "copies, one halo row, one stream": ("copy", {"ISING_RING_GHOST": "1", "ISING_RING_INLINE": "1", "ISING_RING_STORE": "0"}),
}


@pytest.mark.parametrize("case", list(_RING_OF_ONE))
def test_ring_of_one_at_the_bench_size_every_schedule(gpu, monkeypatch, case):
    """One GPU of a ring at the size bench.py runs (65536^2, T_c, seed 1234), through every schedule, transport and buffer
    owner of the slab ring, against the oracle's counts after 5, 21 and 25 sweeps.  At this size the edge-row launch on the
    comm stream really runs next to the interior launch on the compute stream -- which is where the two launches once
    shared their accept-mask slots (wrong spins at 65536^2, right ones at every test size)."""
    kind, env = _RING_OF_ONE[case]
    for k in ("ISING_RING_GHOST", "ISING_RING_INLINE", "ISING_RING_STORE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fx = json.load(open(os.path.join(GOLD, "bench_65536_tc.json")))
    gold = {p["sweeps"]: (p["up"], p["down"]) for p in fx["points"]}
    X, Y, seed = fx["X"], fx["Ytot"], fx["seed"]
    if kind == "native-torch":
        import torch  # noqa: F401
        slab = ig.HipSlabBackend.create(X, Y, device=0, seed=seed, temp=ig.CRIT_TEMP_F32, nslabs=1, slab=0, ring_halo=True).slab
    else:
        slab = ig.IsingSlab(X, Y, seed=seed, temp=ig.CRIT_TEMP_F32, ring_halo=True)
    try:
        ring = ig.NativeRing(slab) if kind.startswith("native") else ig.SlabSet([slab])
        ring.init()
        done = 0
        for upto in (5, 21, 25):
            ring.sweep(upto - done)
            done = upto
            assert ring.count() == gold[upto], (case, upto)
        if kind.startswith("native"):
            ring.close()
    finally:
        slab.close()


def test_lattices_beyond_2_to_the_32_threads_and_a_million_columns(gpu):
    """2^37 spins (16 GiB on the device) in rows of 2^20 columns: the init kernels' grids fold into two dimensions (HIP carries
    2^32 threads per grid dimension; at 2^39 spins they need that many), and the three update forms -- fused ballot launches,
    one ballot launch per colour (what ising_create picks for rows this wide), the dense kernel -- agree on counts and bond sum."""
    import os
    X, Y = 1 << 20, 1 << 17
    res = {}
    for name, lay, env in (("fused", ig.LAYOUT_BALLOT, "1"), ("per colour", ig.LAYOUT_BALLOT, None), ("dense", ig.LAYOUT_DENSE, None)):
        old = os.environ.pop("ISING_FUSED", None)
        if env:
            os.environ["ISING_FUSED"] = env
        try:
            with ig.IsingSlab(X, Y, seed=4321, temp=ig.CRIT_TEMP_F32, layout=lay) as s:
                assert s.fused == (name == "fused")
                s.init()
                c0 = s.count()
                s.sweep(2)
                res[name] = (c0, s.count(), s.bond_equal())
        finally:
            os.environ.pop("ISING_FUSED", None)
            if old is not None:
                os.environ["ISING_FUSED"] = old
    assert res["fused"] == res["per colour"] == res["dense"]
    # ... and with the pinned CPU oracle, streamed over the lattice in chunks of rows (tests/golden/make_golden_huge.py)
    gold = json.load(open(os.path.join(GOLD, "huge_1048576x131072.json")))
    assert (gold["X"], gold["Ytot"], gold["seed"]) == (X, Y, 4321)
    p0, p2 = gold["points"]
    assert p0["sweeps"] == 0 and p2["sweeps"] == 2
    assert res["dense"] == ((p0["up"], p0["down"]), (p2["up"], p2["down"]), p2["bond_equal"])


def test_spin_flip_symmetry_at_the_bench_size(gpu):
    """A property that needs no oracle, at BASELINE config 3's size: the plain model is symmetric under flipping every spin, and the
    random numbers do not depend on the state -- so the run that starts from the complemented lattice (written through the
    1-bit host format, 2 x 256 MiB) is the complement of the run that starts from the lattice itself, bit for bit."""
    X = Y = 65536
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init()
        start = [s.read_bits(c) for c in (ig.BLACK, ig.WHITE)]
        s.sweep(8)
        want = [~s.read_bits(c) for c in (ig.BLACK, ig.WHITE)]
        s.init()
        for c in (ig.BLACK, ig.WHITE):
            s.write_bits(c, ~start[c])
        del start
        up, down = s.count()
        s.sweep(8)
        for c in (ig.BLACK, ig.WHITE):
            assert np.array_equal(s.read_bits(c), want[c]), c
    gold = {p["sweeps"]: p for p in _gold("bench_65536_tc.json")["points"]}[0]
    assert (up, down) == (gold["down"], gold["up"])  # the complemented start, counted


def test_checkpoint_round_trip_at_the_bench_size(gpu, tmp_path):
    """Save -> go on -> load -> go on again at 65536^2 (a 512 MiB file at 1 bit per spin, offsets past 2^31): the continuation repeats
    itself and both meet the oracle's golden counts; the file loads into two slabs as well (it is decomposition independent)."""
    pts = {p["sweeps"]: p for p in _gold("bench_65536_tc.json")["points"]}
    path = str(tmp_path / "bench.ckpt")
    with ig.IsingSlab(65536, 65536, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        one = ig.SlabSet([s]).init()
        one.sweep(2)
        assert one.count() == (pts[2]["up"], pts[2]["down"])
        one.checkpoint_save(path)
        assert os.path.getsize(path) >= 65536 * 65536 // 8
        one.sweep(3)
        assert one.count() == (pts[5]["up"], pts[5]["down"]) and one.bond_equal() == pts[5]["bond_equal"]
        one.checkpoint_load(path)
        assert one.it == 2 and one.count() == (pts[2]["up"], pts[2]["down"])
        one.sweep(3)
        assert one.count() == (pts[5]["up"], pts[5]["down"]) and one.bond_equal() == pts[5]["bond_equal"]
    slabs = [ig.IsingSlab(65536, 32768, seed=1234, temp=ig.CRIT_TEMP_F32, nslabs=2, slab=k) for k in range(2)]
    try:
        two = ig.SlabSet(slabs).init()
        two.checkpoint_load(path)
        assert two.it == 2
        two.sweep(3)
        assert two.count() == (pts[5]["up"], pts[5]["down"]) and two.bond_equal() == pts[5]["bond_equal"]
    finally:
        for s in slabs:
            s.close()


def test_correlation_distance_one_is_the_bond_sum_at_the_bench_size(gpu):
    """-c at 65536^2: the distance-1 entry of the correlation sums is (parallel - antiparallel bonds) = 2 A - 2 N, with A the bond sum
    the oracle's golden holds after 5 sweeps -- the correlation kernel against the oracle at full size through an exact identity."""
    pts = {p["sweeps"]: p for p in _gold("bench_65536_tc.json")["points"]}
    n = 65536 * 65536
    with ig.IsingSlab(65536, 65536, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init().sweep(5)
        sums = s.correlations(2)
        assert sums[0] == 2 * pts[5]["bond_equal"] - 2 * n
        assert abs(sums[1]) < abs(sums[0])  # (distance 2 is less correlated than distance 1 at T_c after 5 sweeps from a hot start)


def test_coupled_update_at_the_bench_size_by_the_antiferromagnetic_map(gpu):
    """-J at 65536^2 (two coupling arrays of 1 GiB): every bond antiferromagnetic and the black colour complemented is the plain run with
    the black colour complemented (tests/test_gpu_couplings.py at 2^25 spins) -- here against the oracle's golden counts after 5 sweeps:
    complementing black turns its up spins into down spins."""
    pts = {p["sweeps"]: p for p in _gold("bench_65536_tc.json")["points"]}
    X = Y = 65536
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32) as s:
        s.init().sweep(5)
        assert s.count() == (pts[5]["up"], pts[5]["down"])
    with ig.IsingSlab(X, Y, seed=1234, temp=ig.CRIT_TEMP_F32, J_prob=1.0) as s:
        s.init()
        full = np.full((Y, X // 32), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        s.write_couplings(ig.BLACK, full)
        s.write_couplings(ig.WHITE, full)
        del full
        b = s.read_bits(ig.BLACK)
        s.write_bits(ig.BLACK, ~b)
        del b
        s.sweep(5)
        # white as in the plain run, black complemented: up_total = up_white + (N/2 - up_black)
        bits = s.read_bits(ig.BLACK)
        up_black_c = sum(int(np.unpackbits(bits[r0:r0 + 4096].view(np.uint8)).sum()) for r0 in range(0, Y, 4096))
        del bits
        up, down = s.count()
        half = X * Y // 2
        up_white = up - up_black_c
        assert (up_white + (half - up_black_c), half - up_white + up_black_c) == (pts[5]["up"], pts[5]["down"])
