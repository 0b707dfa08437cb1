"""Batches of quad-path lattices (round 6; ising_batch.cpp: batch_sweep_quad, ising_quad.hip: QuadRec): ONE quad_pass_k launch per pass carries the tiles of every
member and the draws of every member for the pass to come.  Every member must end up bit for bit where a run of its own ends up (= the oracle) -- whatever the
batch's tile height, pass length and waves, whatever the members' temperatures and seeds --, and the print points that ride in the passes must be every
member's own counts and bond sums."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu

TC = ig.CRIT_TEMP_F32
KEYS = ("ISING_QUAD", "ISING_QUAD_C", "ISING_QUAD_T", "ISING_QUAD_WAVES", "ISING_TILES")


def _env(monkeypatch, **kw):
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in kw.items():
        monkeypatch.setenv("ISING_" + k, str(v))


def _members(X, Y, temps, seeds):
    slabs = [ig.IsingSlab(X, Y, seed=s, temp=t, layout=ig.LAYOUT_DENSE) for t, s in zip(temps, seeds)]
    for s in slabs:
        assert s.quad
    return slabs


def _same(slabs, orcs, what):
    for r, (s, o) in enumerate(zip(slabs, orcs)):
        for color, ref in ((ig.BLACK, o.black), (ig.WHITE, o.white)):
            got = s.read(color)
            if not np.array_equal(got, ref):
                bad = np.argwhere(got != ref)
                raise AssertionError(f"{what}: lattice {r}, colour {color} differs in {len(bad)} words; first at row {bad[0][0]} word {bad[0][1]}")


SHAPES = [  # X, Y, lattices, row groups per tile, sweeps per pass, waves
    (2048, 64, 3, 8, 8, 4), (2048, 32, 5, 4, 4, 8), (2048, 16, 7, 4, 2, 2), (2048, 128, 2, 16, 3, 8), (2048, 48, 4, 5, 6, 12),
    (4096, 64, 3, 8, 5, 8), (4096, 128, 2, 7, 7, 16), (6144, 48, 3, 3, 6, 16), (8192, 64, 2, 4, 4, 8), (12288, 32, 3, 2, 2, 12), (16384, 48, 2, 2, 4, 16),
]


@pytest.mark.parametrize("X,Y,n,C,T,NW", SHAPES)
def test_quad_batch_members_equal_the_oracle(gpu, oracle_mod, monkeypatch, X, Y, n, C, T, NW):
    _env(monkeypatch, QUAD=1, QUAD_C=C, QUAD_T=T, QUAD_WAVES=NW)
    temps = [1.5 + 0.37 * r for r in range(n)]
    seeds = [1000 + 17 * r for r in range(n)]
    slabs = _members(X, Y, temps, seeds)
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
    with ig.IsingBatch(slabs) as b:
        assert b.quad_shape == (min(C, Y // 4), T, NW)
        b.init()
        done = 0
        for upto in (1, 3, 3 + 2 * T, 4 + 4 * T + 1, 4 + 4 * T + 1 + 5 * T + 1):  # a lone sweep; one short pass; exactly two full passes; uneven; many
            b.sweep(upto - done)
            for o in orcs:
                o.sweep(upto - done)
            done = upto
            _same(slabs, orcs, f"after {upto} sweeps ({n} lattices, tiles of {C} row groups, {T} sweeps a pass, {NW} waves)")
        b.measure_enqueue().sweep(2).measure_enqueue()
        meas = b.measure_fetch()
        assert len(meas) == 2
        for r, o in enumerate(orcs):
            assert meas[0][r] == (*o.count(), o.bond_equal())
            o.sweep(2)
            assert meas[1][r] == (*o.count(), o.bond_equal())
    for s in slabs:
        s.close()


@pytest.mark.parametrize("X,Y,n,first,nsw,every", [(2048, 64, 3, 0, 40, 16), (2048, 128, 2, 5, 37, 7), (4096, 64, 4, 0, 9, 1), (2048, 32, 6, 3, 50, 100), (6144, 48, 2, 0, 33, 16)])
@pytest.mark.parametrize("energy", [False, True])
def test_quad_batch_counted_sweeps(gpu, oracle_mod, monkeypatch, X, Y, n, first, nsw, every, energy):
    """ising_batch_sweep_counted: the reference's print points (optimized/main.cu:1785-1798) for every lattice of the batch, inside the passes."""
    _env(monkeypatch, QUAD=1)
    temps = [TC - 0.2 + 0.1 * r for r in range(n)]
    seeds = [77 + (r % 2) for r in range(n)]  # (chains: lattices that share a seed differ in temperature)
    slabs = _members(X, Y, temps, seeds)
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
    with ig.IsingBatch(slabs) as b:
        b.init().sweep(first)
        for o in orcs:
            o.sweep(first)
        got = b.sweep_counted(nsw, every, energy)
        want = []
        for _ in range(nsw):
            for o in orcs:
                o.sweep(1)
            if orcs[0].it % every == 0:
                want.append([o.count() + ((o.bond_equal(),) if energy else (None,)) for o in orcs])
        assert got == want
        _same(slabs, orcs, "after the counted sweeps")
        b.sweep(3)
        for o in orcs:
            o.sweep(3)
        _same(slabs, orcs, "three sweeps later")
    for s in slabs:
        s.close()


def test_quad_batch_default_shape_and_lone_sweeps_interleave(gpu, oracle_mod, monkeypatch):
    """The library's own choice of shape for many tiles; members stay ordinary contexts: swept alone between two batch calls, their temperature changed in
    between (the batch picks the new thresholds up)."""
    _env(monkeypatch, QUAD=1)
    X, Y, n = 2048, 256, 4
    temps, seeds = [2.0, 2.2, 2.4, 2.6], [5, 6, 7, 8]
    slabs = _members(X, Y, temps, seeds)
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
    with ig.IsingBatch(slabs) as b:
        assert b.quad_shape is not None and b.quad_shape[1] <= slabs[0].max_sweeps_per_launch
        b.init().sweep(11)
        for o in orcs:
            o.sweep(11)
        slabs[2].it = 11
        slabs[2].sweep(4)  # alone: its own passes
        orcs[2].sweep(4)
        for r in (0, 1, 3):
            slabs[r].it = 11
            slabs[r].sweep(4)
            orcs[r].sweep(4)
        b.it = 15
        for s, o, t in zip(slabs, orcs, (2.3, 1.7, 2.9, TC)):
            s.set_temperature(t)
            o.temp = float(np.float32(t))
        b.sweep(21)
        for o in orcs:
            o.sweep(21)
        _same(slabs, orcs, "after lone sweeps and a temperature change")
    for s in slabs:
        s.close()


def test_quad_batch_refuses_mixed_kinds(gpu, monkeypatch):
    _env(monkeypatch, QUAD=1)
    a = ig.IsingSlab(8192, 64, seed=1, temp=2.0, layout=ig.LAYOUT_DENSE)
    b = ig.IsingSlab(8192, 64, seed=2, temp=2.0, layout=ig.LAYOUT_BALLOT)
    assert a.quad and not b.quad
    with pytest.raises(ig.IsingError, match="same way"):
        ig.IsingBatch([a, b])
    with pytest.raises(ig.IsingError, match="same way"):
        ig.IsingBatch([b, a])
    a.close()
    b.close()


def test_quad_batch_at_a_temperature_series_size(gpu, oracle_mod, monkeypatch):
    """31 lattices of 2048 x 512 (a temperature series' worth of tiles at a size the oracle finishes in seconds), the library's shape, print points with the energy."""
    _env(monkeypatch)
    X, Y, n = 2048, 512, 31
    temps = [1.5 + 0.05 * r for r in range(n)]
    slabs = [ig.IsingSlab(X, Y, seed=ig.SEED_DEF, temp=t) for t in temps]
    if not slabs[0].quad:
        pytest.skip("the quad path is the default on a whole MI355X only")
    orcs = [oracle_mod.OracleLattice(X, Y, seed=ig.SEED_DEF, temp=t).init() for t in temps]
    with ig.IsingBatch(slabs) as b:
        b.init()
        got = b.sweep_counted(24, 8, True)
        want = []
        for _ in range(3):
            for o in orcs:
                o.sweep(8)
            want.append([o.count() + (o.bond_equal(),) for o in orcs])
        assert got == want
        _same(slabs, orcs, "31 lattices after 24 sweeps")
    for s in slabs:
        s.close()


def test_quad_batch_randomised(gpu, oracle_mod, monkeypatch):
    """Random batches -- lattice widths, rows, lattices per batch, tile heights, halo depths, workgroup sizes, call lengths, print points with and without the energy --
    against the oracle, every member (seeded: the same cases every run)."""
    rng = np.random.default_rng(20260931)
    done = 0
    for case in range(48):
        gx = int(rng.integers(1, 5))
        X = 2048 * gx
        Y = 16 * int(rng.integers(1, 9))
        n = int(rng.integers(2, 7))
        T = int(rng.integers(1, 9 if gx > 1 else 13))
        C = int(rng.integers(1, 9 if gx > 1 else 17))
        NW = int(rng.choice([4, 8, 12, 16] if gx > 1 else [1, 2, 4, 8, 12, 16]))
        HG = (2 * T + 2) // 4
        items = (min(C, Y // 4) + 2 * HG) * gx
        per_wave = (items + NW - 1) // NW
        if per_wave > 4 or (per_wave > 3 and NW > 8) or items * 1024 > 150 * 1024:
            continue
        temps = [float(rng.choice([1.5, 2.0, TC, 2.5, 3.0])) for _ in range(n)]
        seeds = [int(rng.integers(1, 2**62)) for _ in range(n)]
        _env(monkeypatch, QUAD=1, QUAD_C=C, QUAD_T=T, QUAD_WAVES=NW)
        slabs = _members(X, Y, temps, seeds)
        orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
        with ig.IsingBatch(slabs) as b:
            b.init()
            for k in rng.integers(1, 4 * T + 4, size=3):
                k = int(k)
                if rng.integers(0, 2) == 0:
                    every, energy = int(rng.integers(1, 9)), bool(rng.integers(0, 2))
                    got = b.sweep_counted(k, every, energy)
                    want = []
                    for _ in range(k):
                        for o in orcs:
                            o.sweep(1)
                        if orcs[0].it % every == 0:
                            want.append([o.count() + ((o.bond_equal(),) if energy else (None,)) for o in orcs])
                    assert got == want, (case, X, Y, n, C, T, NW)
                else:
                    b.sweep(k)
                    for o in orcs:
                        o.sweep(k)
                _same(slabs, orcs, f"case {case}: {n} x {Y} x {X}, tiles of {C} row groups, {T} sweeps a pass, {NW} waves, after {orcs[0].it} sweeps")
        for s in slabs:
            s.close()
        done += 1
    assert done >= 16, done
