"""Ballot layout on lattices whose width is not a multiple of 8192 columns: the last wave column of every row is partly
dead (padded rows, dead lanes stay zero).  Everything that touches the row geometry against the oracle: initialisation,
plain and fused updates incl. the periodic wrap through the partial wave column, counts, bond sums, correlations, the
boundary formats, the C-ABI ring, and the re-layout when the slab has to turn dense."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu


def _same(s, orc):
    return np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)


@pytest.mark.parametrize("X", [2048, 4096, 6144, 10240, 12288, 14336, 22528])
@pytest.mark.parametrize("fused", ["0", "1"])
def test_partial_wave_column_matches_oracle(gpu, oracle_mod, monkeypatch, X, fused):
    monkeypatch.setenv("ISING_FUSED", fused)
    Y, seed, temp = 48, 77, ig.CRIT_TEMP_F32
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_BALLOT) as s:
        assert s.layout == ig.LAYOUT_BALLOT
        s.init()
        assert _same(s, orc) and s.count() == orc.count()
        for n in (1, 2, 5):
            s.sweep(n)
            orc.sweep(n)
            assert _same(s, orc), (X, s.it)
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()
        assert s.correlations(16) == orc.corr(16)


def test_partial_wave_column_formats_ring_and_relayout(gpu, oracle_mod):
    X, Yk, n, seed = 10240, 32, 3, 9
    orc = oracle_mod.OracleLattice(X, Yk * n, seed=seed, temp=2.0).init().sweep(4)
    ring = ig.SlabSet([ig.IsingSlab(X, Yk, seed=seed, temp=2.0, nslabs=n, slab=k, layout=ig.LAYOUT_BALLOT) for k in range(n)])
    try:
        ring.init().sweep(4)
        assert ring.count() == orc.count() and ring.bond_equal() == orc.bond_equal()
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in ring.slabs]), orc.black)
        # a temperature without integer thresholds: every slab turns dense (shorter rows in the same buffer) and goes on
        ring.set_temperature(-1.0)
        ring.sweep(2)
        orc.temp = -1.0
        orc.sweep(2)
        assert all(s.current_layout() == ig.LAYOUT_DENSE for s in ring.slabs)
        assert ring.count() == orc.count()
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in ring.slabs]), orc.white)
    finally:
        ring.close()
    # write the oracle's rows into a fresh slab (1 bit/spin and packed), continue, compare
    ref = oracle_mod.OracleLattice(X, 64, seed=seed, temp=2.0).init().sweep(1)
    with ig.IsingSlab(X, 64, seed=seed, temp=2.0, layout=ig.LAYOUT_BALLOT) as s:
        s.write(ig.BLACK, ref.black)
        s.write(ig.WHITE, ref.white)
        s.it = 1
        s.sweep(2)
        ref.sweep(2)
        assert _same(s, ref)
        bits = s.read_bits(ig.BLACK)
        assert bits.shape == (64, X // 64) and int(np.unpackbits(bits.view(np.uint8)).sum()) == int(np.unpackbits(ref.black.view(np.uint8)).sum())


def test_auto_layout_and_partial_wave_columns(gpu):
    with ig.IsingSlab(2048 * 27, 4096, temp=2.0) as s:  # 2^27.8 spins, 27 column groups: 6 full wave columns + 3 of 4 groups
        assert s.layout == ig.LAYOUT_BALLOT
    with ig.IsingSlab(2048 * 13, 8192, temp=2.0) as s:  # 13 groups of 16: 19 % of dead lanes still pay (end of round 4: a sixth pays at every size)
        assert s.layout == ig.LAYOUT_BALLOT
    with ig.IsingSlab(2048 * 9, 8192, temp=2.0) as s:   # 9 groups of 12: a quarter of dead lanes above 2^27 spins -> dense pays
        assert s.layout == ig.LAYOUT_DENSE
    with ig.IsingSlab(2048 * 27, 4096, temp=2.0, XSL=2048, YSL=2048) as s:
        assert s.layout == ig.LAYOUT_DENSE  # sub-lattices of other widths stay on the dense kernel
