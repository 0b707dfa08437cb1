"""Batched fused launches (ising_batch_*): independent lattices of one shape share the tickets of one launch.  Every member
must end up bit for bit where a run of its own ends up (= the oracle), whatever the batch's strip height and grid; the
batch's one-launch measurement must give ising_count / ising_bond_equal of every member."""
import os

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu


def _members(X, Y, temps, seeds, **kw):
    return [ig.IsingSlab(X, Y, seed=s, temp=t, layout=ig.LAYOUT_BALLOT, **kw) for t, s in zip(temps, seeds)]


@pytest.mark.parametrize("X,Y,n", [(8192, 128, 3), (8192, 64, 5), (16384, 32, 2), (10240, 48, 3), (32768, 16, 4)])
def test_batch_members_equal_the_oracle(gpu, oracle_mod, X, Y, n):
    temps = [1.5 + 0.4 * r for r in range(n)]
    seeds = [1000 + 17 * r for r in range(n)]
    slabs = _members(X, Y, temps, seeds)
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip(temps, seeds)]
    with ig.IsingBatch(slabs) as b:
        b.init()
        for k in (1, 3, 37):  # one launch, one launch, two launches (32 + 5 sweeps)
            b.sweep(k).measure_enqueue()
            for o in orcs:
                o.sweep(k)
            for s, o in zip(slabs, orcs):
                assert np.array_equal(s.read(ig.BLACK), o.black) and np.array_equal(s.read(ig.WHITE), o.white)
        meas = b.measure_fetch()
        assert len(meas) == 3
        for r, o in enumerate(orcs):  # the last measurement against the oracle, every measurement's count against the slab's own
            assert meas[-1][r] == (*o.count(), o.bond_equal())
            assert slabs[r].count() == o.count() and slabs[r].bond_equal() == o.bond_equal()
    for s in slabs:
        s.close()


def test_batch_and_lone_sweeps_interleave(gpu, oracle_mod):
    """Members stay ordinary contexts: swept alone between two batch sweeps (their own tickets and counters), temperature
    changed in between (the batch picks the new thresholds up)."""
    X, Y = 8192, 256
    slabs = _members(X, Y, [2.0, 2.5], [7, 8])
    orcs = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init() for t, s in zip([2.0, 2.5], [7, 8])]
    with ig.IsingBatch(slabs) as b:
        b.init().sweep(4)
        for o in orcs:
            o.sweep(4)
        slabs[1].it = 4
        slabs[1].sweep(2)   # alone
        orcs[1].sweep(2)
        slabs[0].it = 4
        slabs[0].sweep(2)
        orcs[0].sweep(2)
        b.it = 6
        for s, o, t in zip(slabs, orcs, (2.3, 1.7)):
            s.set_temperature(t)
            o.temp = float(np.float32(t))
        b.sweep(5)
        for o in orcs:
            o.sweep(5)
        for s, o in zip(slabs, orcs):
            assert np.array_equal(s.read(ig.BLACK), o.black) and np.array_equal(s.read(ig.WHITE), o.white)
    for s in slabs:
        s.close()


def test_batch_shapes_with_forced_grids(gpu, oracle_mod, monkeypatch):
    """Small persistent grids (tickets of several lattices interleave on few workgroups) and the full one."""
    X, Y, n = 8192, 64, 4
    temps, seeds = [2.0, 2.2, 2.4, 2.6], [5, 6, 7, 8]
    ref = [oracle_mod.OracleLattice(X, Y, seed=s, temp=t).init().sweep(9) for t, s in zip(temps, seeds)]
    for wgs in ("1", "3", "64", None):
        if wgs is None:
            monkeypatch.delenv("ISING_FUSED_WGS", raising=False)
        else:
            monkeypatch.setenv("ISING_FUSED_WGS", wgs)
        slabs = _members(X, Y, temps, seeds)
        with ig.IsingBatch(slabs) as b:
            b.init().sweep(9)
            for s, o in zip(slabs, ref):
                assert np.array_equal(s.read(ig.BLACK), o.black) and np.array_equal(s.read(ig.WHITE), o.white), wgs
        for s in slabs:
            s.close()


def test_batch_rejects_what_it_cannot_carry(gpu):
    a = ig.IsingSlab(8192, 64, layout=ig.LAYOUT_BALLOT)
    d = ig.IsingSlab(8192, 64, layout=ig.LAYOUT_DENSE)
    o = ig.IsingSlab(8192, 128, layout=ig.LAYOUT_BALLOT)
    r = ig.IsingSlab(8192, 64, layout=ig.LAYOUT_BALLOT, nslabs=2, slab=0)
    for bad in ([a, d], [a, o], [a, r], [a, a]):
        with pytest.raises(ig.IsingError):
            ig.IsingBatch(bad)
    # a temperature without integer thresholds: the batch refuses the sweep instead of sweeping with stale ones
    with ig.IsingBatch([a]) as b:
        b.init().sweep(1)
        a.set_temperature(0.0)
        with pytest.raises(ig.IsingError):
            b.sweep(1)
    for s in (a, d, o, r):
        s.close()


def test_batch_at_config5_size_counts(gpu):
    """BASELINE config 5's lattices: 8192^2, a handful of temperatures side by side, against each lattice's own fused run."""
    X = Y = 8192
    temps = [1.5, 2.25, 3.0]
    lone = []
    for t in temps:
        with ig.IsingSlab(X, Y, seed=1234, temp=t) as s:
            s.init().sweep(40)
            lone.append((*s.count(), s.bond_equal()))
    slabs = _members(X, Y, temps, [1234] * 3)
    with ig.IsingBatch(slabs) as b:
        assert b.strip_rows >= 2
        b.init().sweep(40).measure_enqueue()
        assert b.measure_fetch()[0] == lone
    for s in slabs:
        s.close()
