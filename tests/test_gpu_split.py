"""GPU parity of the split launches (ising_ballot.hip: ballot_split_k -- draw units and word units with tickets of their own, the accept masks through
a ring per XCD) against the CPU oracle, bit for bit: every word of both colours, counts, bond sum, the print points inside the launches -- over strip
heights, grids (more workgroups than tickets, fewer than classes want), leads, partly dead wave columns, launches of many levels."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
TC = ig.CRIT_TEMP_F32


def _same(slab, orc):
    return np.array_equal(slab.read(ig.BLACK), orc.black) and np.array_equal(slab.read(ig.WHITE), orc.white)


def _env(monkeypatch, lead=None, wgs=None):
    monkeypatch.setenv("ISING_SPLIT", "1")
    monkeypatch.setenv("ISING_FUSED", "1")
    for k, v in (("ISING_SPLIT_LEAD", lead), ("ISING_FUSED_WGS", wgs)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, str(v))


@pytest.mark.parametrize("X,Y,H,lead,wgs,sweeps", [
    (8192, 64, 1, 1, None, (1, 2, 5)), (8192, 128, 4, 0, None, (3, 7)), (8192, 256, 8, 1, 64, (2, 9)), (16384, 256, 16, 1, None, (4, 4)),
    (16384, 512, 16, 2, 512, (11,)), (24576, 128, 4, 1, None, (3, 3)), (10240, 128, 2, 1, None, (5, 2)), (65536, 64, 4, 1, 256, (6,)),
    (8192, 2048, 16, 1, None, (33,)), (8192, 1024, 2, 0, 1536, (40,)),
])
def test_split_launches_bit_exact(gpu, oracle_mod, monkeypatch, X, Y, H, lead, wgs, sweeps):
    _env(monkeypatch, lead, wgs)
    orc = oracle_mod.OracleLattice(X, Y, seed=4711, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=4711, temp=TC, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        assert s.fused and s.split and s.strip_rows == H
        s.init()
        for n in sweeps:
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc), (X, Y, H, n)
        assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


@pytest.mark.parametrize("X,Y,H,every,calls", [(8192, 128, 4, 3, (7, 2, 70)), (16384, 256, 16, 16, (40, 100)), (10240, 64, 2, 4, (9, 130)), (8192, 512, 8, 1, (5, 3))])
@pytest.mark.parametrize("energy", [False, True], ids=["counts", "counts+energy"])
def test_split_launches_print_points(gpu, oracle_mod, monkeypatch, X, Y, H, every, calls, energy):
    """ising_sweep_counted on split launches: the word units count what they store (and the white levels' equal bonds)"""
    _env(monkeypatch)
    orc = oracle_mod.OracleLattice(X, Y, seed=77, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=77, temp=TC, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        assert s.split
        s.init()
        for n in calls:
            got = s.sweep_counted(n, every, energy)
            want = []
            for _ in range(n):
                orc.sweep(1)
                if orc.it % every == 0:
                    want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
            assert got == want, (n, s.it)
        assert _same(s, orc)


def test_split_launches_low_temperature_and_long_launch(gpu, oracle_mod, monkeypatch):
    """T = 1.5 (other thresholds), one launch of 600 sweeps = 1200 levels: the rings wrap many times"""
    _env(monkeypatch)
    X, Y = 8192, 128
    orc = oracle_mod.OracleLattice(X, Y, seed=ig.SEED_DEF, temp=1.5).init().sweep(600)
    with ig.IsingSlab(X, Y, seed=ig.SEED_DEF, temp=1.5, layout=ig.LAYOUT_BALLOT, strip_rows=2) as s:
        s.init().sweep(600)
        assert _same(s, orc)


@pytest.mark.parametrize("X,Y,H,lead,wgs,sweeps", [
    (8192, 176, 4, 1, None, (3, 4)), (8192, 720, 16, 1, None, (5,)), (12288, 80, 4, 1, 96, (2, 7)), (20480, 80, 2, 0, None, (6,)),
    (8192, 2032, 8, 1, None, (9,)), (16384, 1008, 8, 2, None, (4, 4)),
])
def test_split_launches_any_ticket_count(gpu, oracle_mod, monkeypatch, X, Y, H, lead, wgs, sweeps):
    """levels whose tickets are not a multiple of eight (11, 12, 10, 30, 64 with a workgroup whose last waves have no unit, 63)"""
    _env(monkeypatch, lead, wgs)
    orc = oracle_mod.OracleLattice(X, Y, seed=15, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=15, temp=TC, layout=ig.LAYOUT_BALLOT, strip_rows=H) as s:
        assert s.fused and s.split
        s.init()
        for n in sweeps:
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc), (X, Y, H, n)


@pytest.mark.parametrize("nslabs,Yk,H", [(2, 256, 8), (3, 128, 4), (2, 384, 16), (4, 128, 8)])
def test_split_ring_slabs_one_device(gpu, oracle_mod, monkeypatch, nslabs, Yk, H):
    """ring slabs with ghost rows in the split form (global rows around the ring, the trapezoid, launches that take turns with copies of the ghost rows):
    the whole lattice against the oracle, more sweeps than one exchange period"""
    _env(monkeypatch)
    X, seed, sweeps = 8192, 77, (3, 40, 33)
    orc = oracle_mod.OracleLattice(X, Yk * nslabs, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=TC, nslabs=nslabs, slab=k, layout=ig.LAYOUT_BALLOT, strip_rows=H) for k in range(nslabs)])
    try:
        ring.init()
        assert all(s.split for s in ring.slabs)
        for n in sweeps:
            ring.sweep(n)
            orc.sweep(n)
            assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black), n
            assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), orc.white), n
        assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal()
    finally:
        ring.close()


@pytest.mark.parametrize("every,energy", [(16, False), (5, True)])
def test_split_ring_of_one_overlapped_and_counted(gpu, oracle_mod, monkeypatch, every, energy):
    """the library's rank ring with one rank (RCCL to itself, the exchange next to the launches: edge units, edge_go / edge_done) in the split form, print
    points inside the launches"""
    import torch  # noqa: F401
    _env(monkeypatch)
    X, Y, seed = 8192, 512, 9
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=oracle_mod.CRIT_TEMP).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=TC, layout=ig.LAYOUT_BALLOT, strip_rows=8, ring_halo=True) as s:
        ring = ig.NativeRing(s).init()
        assert s.split
        for n in (70, 45):
            got = ring.sweep_counted(n, every, energy)
            want = []
            for _ in range(n):
                orc.sweep(1)
                if orc.it % every == 0:
                    want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
            assert [tuple(g) for g in got] == want
        ring.quiesce()
        assert _same(s, orc)
        ring.close()


def test_auto_short_calls_fused_long_calls_split(gpu, monkeypatch):
    """ising_create's own choice (no ISING_SPLIT): a lattice whose long calls run split launches keeps the fused form's shape for the short ones (a split launch
    costs ~0.14 ms more than a fused one).  Calls of both kinds in turn -- each form with completion counters of its own -- against the same calls with
    ISING_SPLIT=0: every word, the series of the print points (both forms are held against the oracle on their own, above and in test_gpu_fused.py)."""
    X, Y = 32768, 2048
    calls = [("sweep", 40), ("sweep", 600), ("sweep", 7), ("counted", 520, 16, True), ("counted", 30, 4, False), ("sweep", 512), ("sweep", 1)]

    def run(split_env):
        if split_env is None:
            monkeypatch.delenv("ISING_SPLIT", raising=False)
        else:
            monkeypatch.setenv("ISING_SPLIT", split_env)
        out = []
        with ig.IsingSlab(X, Y, seed=31, temp=TC) as s:
            shape = (s.split, s.strip_rows)
            s.init()
            for c in calls:
                if c[0] == "sweep":
                    s.sweep(c[1])
                else:
                    out.append(s.sweep_counted(c[1], c[2], c[3]))
            out += [s.read(ig.BLACK), s.read(ig.WHITE), s.count(), s.bond_equal()]
        return shape, out

    shape_auto, got = run(None)
    shape_off, want = run("0")
    assert shape_auto[0] and not shape_off[0] and shape_auto[1] > shape_off[1], (shape_auto, shape_off)
    assert got[0] == want[0] and got[1] == want[1]
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]) and got[4:] == want[4:]
