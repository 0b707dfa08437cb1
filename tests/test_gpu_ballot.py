"""The ballot device layout (ising_ballot.hip: 1 bit per spin, row bits in the update kernel's wave-ballot order, accept
decisions handed over as v_cmp lane masks through scalar stores) against the CPU oracle, bit for bit.

It needs X % 8192 == 0, so it has its own size table; everything that crosses the C-ABI is still the reference's
packed layout."""
import numpy as np
import pytest

import ising_gpu_amd as ig
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu
TC = ig.CRIT_TEMP_F32
BAL = ig.LAYOUT_BALLOT


@pytest.mark.parametrize("X,Y,strip", [(8192, 16, 0), (8192, 64, 4), (16384, 48, 16), (24576, 32, 1), (8192, 272, 16)])
@pytest.mark.parametrize("temp,seed", [(1.5, ig.SEED_DEF), (TC, 1234)])
def test_state_bit_exact(gpu, oracle_mod, X, Y, strip, temp, seed):
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=strip, layout=BAL) as s:
        assert s.layout == BAL
        s.init()
        _compare(s, orc, "init")
        assert s.count() == orc.count()
        done = 0
        for upto in (1, 2, 9):
            s.sweep(upto - done)
            orc.sweep(upto - done)
            done = upto
            _compare(s, orc, f"after {upto} sweeps (strip {s.strip_rows})")
            assert s.count() == orc.count()
            assert s.bond_equal() == orc.bond_equal()


@pytest.mark.parametrize("X,Y,XSL,YSL,strip", [(8192, 64, 2048, 16, 0), (8192, 96, 4096, 32, 16), (16384, 48, 8192, 48, 0), (16384, 64, 16384, 16, 8),
                                               (32768, 32, 16384, 32, 4), (24576, 48, 8192, 16, 1), (8192, 64, 8192, 32, 0)])
def test_sublattices_bit_exact(gpu, oracle_mod, X, Y, XSL, YSL, strip):
    """--xsl/--ysl: every XSL x YSL block is an independent torus (loadTile wrap arguments, optimized/main.cu:413-459);
    widths of one, two, four and more column groups take different paths through the side-word masks."""
    orc = oracle_mod.OracleLattice(X, Y, seed=8, temp=2.0, XSL=XSL, YSL=YSL).init()
    with ig.IsingSlab(X, Y, seed=8, temp=2.0, XSL=XSL, YSL=YSL, strip_rows=strip, layout=BAL) as s:
        assert s.layout == BAL
        s.init()
        for n in (1, 4):
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"sub-lattices {XSL}x{YSL} after {s.it}")
        assert s.bond_equal() == orc.bond_equal()
        assert np.array_equal(s.correlations(128), orc.corr(128)) if YSL >= 128 else True


def test_sublattice_row_ranges(gpu, oracle_mod):
    """Row ranges that start and end inside sub-lattices (strips straddle the seams)."""
    X, Y, XSL, YSL = 8192, 96, 2048, 32
    orc = oracle_mod.OracleLattice(X, Y, seed=21, temp=1.9, XSL=XSL, YSL=YSL).init()
    with ig.IsingSlab(X, Y, seed=21, temp=1.9, XSL=XSL, YSL=YSL, strip_rows=16, layout=BAL) as s:
        s.init()
        for it in (1, 2):
            for color in (ig.BLACK, ig.WHITE):
                for lo, hi in ((0, 5), (5, 31), (31, 33), (33, 96)):
                    s.update_color(it, color, lo, hi)
            s.it = it
            orc.sweep(1)
            _compare(s, orc, f"it {it}")


@pytest.mark.parametrize("X,Y,prob,sl", [(8192, 32, 0.3, None), (16384, 64, 0.5, None), (8192, 48, 1.0, None), (8192, 16, 0.0, None),
                                          (16384, 64, 0.4, (2048, 32)), (16384, 32, 0.25, (8192, 16))])
def test_couplings_bit_exact(gpu, oracle_mod, X, Y, prob, sl):
    """-J: coupling arrays (read back through the boundary format) and the coupled update, with and without sub-lattices."""
    kw = dict(XSL=sl[0], YSL=sl[1]) if sl else {}
    orc = oracle_mod.OracleLattice(X, Y, seed=1234, temp=1.8, **kw).init().init_couplings(prob)
    with ig.IsingSlab(X, Y, seed=1234, temp=1.8, J_prob=prob, layout=BAL, **kw) as s:
        assert s.layout == BAL
        s.init().init_couplings()
        assert np.array_equal(s.read_couplings(ig.BLACK), orc.hamB)
        assert np.array_equal(s.read_couplings(ig.WHITE), orc.hamW)
        for n in (1, 5):
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"-J {prob} after {s.it}")


def test_couplings_ring_and_migration(gpu, oracle_mod):
    """-J on a ring of ballot slabs (the white couplings need the neighbours' black edge rows), then a temperature
    without integer thresholds: spins and couplings turn dense together."""
    X, Y, n = 8192, 96, 3
    orc = oracle_mod.OracleLattice(X, Y, seed=4, temp=1.5).init().init_couplings(0.35)
    backs = [ig.HipSlabBackend.create(X, Y // n, seed=4, temp=1.5, nslabs=n, slab=k, J_prob=0.35, layout=BAL) for k in range(n)]
    try:
        ring = ig.LocalRing(backs).init()
        ring.sweep(3)
        orc.sweep(3)
        assert np.array_equal(np.concatenate([b.slab.read_couplings(ig.WHITE) for b in backs]), orc.hamW)
        assert np.array_equal(np.concatenate([b.slab.read(ig.WHITE) for b in backs]), orc.white)
        for b in backs:
            b.slab.set_temperature(0.0)
        orc.temp = 0.0
        ring.sweep(2)
        orc.sweep(2)
        assert all(b.slab.current_layout() == ig.LAYOUT_DENSE for b in backs)
        assert np.array_equal(np.concatenate([b.slab.read_couplings(ig.BLACK) for b in backs]), orc.hamB)
        assert np.array_equal(np.concatenate([b.slab.read(ig.BLACK) for b in backs]), orc.black)
        assert np.array_equal(np.concatenate([b.slab.read(ig.WHITE) for b in backs]), orc.white)
    finally:
        for b in backs:
            b.slab.close()


def test_auto_layout_picks_ballot_where_it_applies(gpu):
    with ig.IsingSlab(16384, 8192, temp=1.5) as s:      # from 2^27 spins per slab up
        assert s.layout == BAL
    with ig.IsingSlab(8192, 8192, temp=1.5) as s:       # from 2^26 spins where ising_sweep's fused launches apply ...
        assert s.layout == BAL and s.fused and s.split and s.strip_rows == 4   # (round 5: the split form where sixteen-row strips make under 2048 tickets a level)
    with ig.IsingSlab(16384, 16384, temp=1.5) as s:     # fused launches all the way up, strip height by tickets per level
        assert s.layout == BAL and s.fused and s.split and s.strip_rows == 16
    with ig.IsingSlab(32768, 32768, temp=1.5) as s:
        assert s.layout == BAL and s.fused and not s.split and s.strip_rows == 8
    with ig.IsingSlab(65536, 1024, temp=1.5) as s:      # (wide and short: too few strips for the split form)
        assert s.layout == BAL and s.fused and not s.split
    with ig.IsingSlab(8192, 8192, temp=1.5, ring_halo=True) as s:   # ... also for a ring slab: ghost rows, fused launches between exchanges
        # (round 6: a ring slab of few tickets has a split shape as well -- the ring uses it in launches of several exchange epochs, ising_ring.cpp)
        assert s.layout == BAL and s.fused and s.split and s.strip_rows == 4 and s.ghost_ptrs(ig.BLACK)[0] == 64  # (the split shape's strips; the fused form's: 2)
    with ig.IsingSlab(8192, 8192, temp=1.5, nslabs=2, J_prob=0.2) as s:   # (with -J too)
        assert s.layout == BAL and s.fused
    with ig.IsingSlab(8192, 4096, temp=1.5) as s:       # ... down to 2^25 spins
        assert s.layout == BAL and s.fused and s.strip_rows == 2  # (round 4: units draw before they wait -- two-row units with 512 tickets a level)
    with ig.IsingSlab(8192, 2048, temp=1.5) as s:       # ... and below, while a level of one-row units still feeds two workgroups per CU (end of round 4)
        assert s.layout == BAL and s.fused and s.strip_rows == 1
    with ig.IsingSlab(16384, 2048, temp=1.5) as s:      # (rows of several wave columns want more tickets a level before two-row units pay)
        assert s.layout == BAL and s.fused and s.strip_rows == 1
    with ig.IsingSlab(65536, 256, temp=1.5) as s:       # small slabs: the dense layout is ahead (tile launches; up to six blocks of 2048 columns: the quad path)
        assert s.layout == ig.LAYOUT_DENSE and s.tiled
    with ig.IsingSlab(8192, 1024, temp=1.5) as s:
        assert s.layout == ig.LAYOUT_DENSE and s.quad
    with ig.IsingSlab(12288, 2048, temp=1.5) as s:      # 1.5 wave columns: up to 2^26 spins a quarter of dead lanes costs less than the dense layout's launches
        assert s.layout == BAL and s.fused
    with ig.IsingSlab(20480, 8192, temp=1.5) as s:      # 10 column groups = 2.5 wave columns: a sixth of dead lanes pays at every size (end of round 4)
        assert s.layout == BAL
    with ig.IsingSlab(12288, 16384, temp=1.5) as s:     # 1.5 wave columns: a quarter of dead lanes pays up to 2^27 spins only
        assert s.layout == ig.LAYOUT_DENSE
    with ig.IsingSlab(16384, 8192, temp=1.5, XSL=2048, YSL=16) as s:
        assert s.layout == BAL
    with ig.IsingSlab(24576, 8192, temp=1.5, XSL=6144, YSL=16) as s:   # sub-lattice width of three column groups
        assert s.layout == ig.LAYOUT_DENSE
    with ig.IsingSlab(16384, 8192, temp=1.5, J_prob=0.1) as s:
        assert s.layout == BAL
    with ig.IsingSlab(16384, 8192, temp=0.0) as s:      # no integer thresholds at T = 0
        assert s.layout == ig.LAYOUT_DENSE
    with ig.IsingSlab(4096, 32, temp=1.5, layout=BAL) as s:   # on request: any width (tests/test_gpu_ballot_partial.py) ...
        assert s.layout == BAL
    with pytest.raises(ig.IsingError):                         # ... but not with sub-lattices or -J on top of a partial wave column
        ig.IsingSlab(4096, 32, temp=1.5, layout=BAL, XSL=2048, YSL=16)
    with pytest.raises(ig.IsingError):
        ig.IsingSlab(4096, 32, temp=1.5, layout=BAL, J_prob=0.2)


def test_row_partition_and_edges(gpu, oracle_mod):
    """Any partition of the rows into launches gives the same state (what the slab ring relies on)."""
    X, Y = 8192, 96
    orc = oracle_mod.OracleLattice(X, Y, seed=5, temp=2.0).init()
    with ig.IsingSlab(X, Y, seed=5, temp=2.0, layout=BAL) as s:
        s.init()
        for it in (1, 2, 3):
            for color in (ig.BLACK, ig.WHITE):
                s.update_edges(it, color)
                s.update_color(it, color, 1, 40)
                s.update_color(it, color, 40, 41)
                s.update_color(it, color, 41, Y - 1)
            s.it = it
            orc.sweep(1)
            _compare(s, orc, f"it {it}")


def test_ring_of_slabs_matches_oracle(gpu, oracle_mod):
    X, Y, n = 8192, 192, 3
    orc = oracle_mod.OracleLattice(X, Y, seed=77, temp=TC).init()
    orc.sweep(6)
    backs = [ig.HipSlabBackend.create(X, Y // n, seed=77, temp=TC, nslabs=n, slab=k, layout=BAL) for k in range(n)]
    try:
        ring = ig.LocalRing(backs).init()
        ring.sweep(6)
        assert np.array_equal(np.concatenate([b.slab.read(ig.BLACK) for b in backs]), orc.black)
        assert np.array_equal(np.concatenate([b.slab.read(ig.WHITE) for b in backs]), orc.white)
        assert ring.count() == orc.count()
        assert ring.bond_equal() == orc.bond_equal()
    finally:
        for b in backs:
            b.slab.close()


def test_write_read_resume_and_correlations(gpu, oracle_mod):
    X, Y = 8192, 256
    orc = oracle_mod.OracleLattice(X, Y, seed=3, temp=2.1).init()
    orc.sweep(3)
    with ig.IsingSlab(X, Y, seed=3, temp=2.1, layout=BAL) as s:
        s.write(ig.BLACK, orc.black)
        s.write(ig.WHITE, orc.white)
        s.it = orc.it
        assert np.array_equal(s.read(ig.BLACK), orc.black)
        s.sweep(2)
        orc.sweep(2)
        _compare(s, orc, "resumed")
        assert np.array_equal(s.correlations(128), orc.corr(128))


def test_temperature_without_thresholds_turns_dense(gpu, oracle_mod):
    """-u style ramp through a temperature the integer thresholds cannot express: the slab migrates to the dense layout."""
    X, Y = 8192, 32
    orc = oracle_mod.OracleLattice(X, Y, seed=11, temp=1.5).init()
    with ig.IsingSlab(X, Y, seed=11, temp=1.5, layout=BAL) as s:
        s.init().sweep(2)
        orc.sweep(2)
        for temp in (0.0, 2.5):
            s.set_temperature(temp)
            orc.temp = float(np.float32(temp))
            s.sweep(2)
            orc.sweep(2)
            _compare(s, orc, f"T={temp}")
        assert s.current_layout() == ig.LAYOUT_DENSE


def test_large_iteration_index_counter_carry(gpu, oracle_mod):
    """From iteration 2^27 on the 64-bit Philox block counter carries into its second word (optimized/main.cu:621)."""
    X, Y, seed, temp = 8192, 32, 55, 2.0
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=BAL) as s:
        s.init()
        for it in (2**27 - 1, 2**27, 2**30 + 12345, 2**31 - 1):
            for color in (ig.BLACK, ig.WHITE):
                s.update_color(it, color)
                orc.update_color(it, color)
            _compare(s, orc, f"it={it}")


def test_wide_lattice_and_empty_ranges(gpu, oracle_mod):
    """X = 262144 (32 wave columns) on the minimum height, empty row ranges, and a strip height that does not divide
    the range (ragged last strip)."""
    X, Y = 262144, 16
    orc = oracle_mod.OracleLattice(X, Y, seed=3, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=3, temp=TC, layout=BAL, strip_rows=8) as s:
        s.init()
        assert s.count() == orc.count()
        s.update_color(1, ig.BLACK, 5, 5)   # empty range
        s.update_color(1, ig.BLACK, 0, 0)
        _compare(s, orc, "after empty launches")
        for color in (ig.BLACK, ig.WHITE):
            s.update_color(1, color, 0, 11)  # strips of 8 + 3 rows
            s.update_color(1, color, 11, 16)
        s.it = 1
        orc.sweep(1)
        _compare(s, orc, "ragged strips")
        s.sweep(2)
        orc.sweep(2)
        _compare(s, orc, "wide lattice")
        assert s.bond_equal() == orc.bond_equal()


def test_ring_correlations_and_cli_layout_flag(gpu, oracle_mod):
    """Correlations over a ring of ballot slabs (the vertical part needs the next slab's first rows), and the CLI's
    --layout flag: identical transcripts for all three layouts."""
    import os
    import subprocess
    X, Y, n = 8192, 384, 3
    orc = oracle_mod.OracleLattice(X, Y, seed=5, temp=2.0).init().sweep(3)
    slabs = [ig.IsingSlab(X, Y // n, seed=5, temp=2.0, nslabs=n, slab=k, layout=BAL) for k in range(n)]
    try:
        ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init().sweep(3)
        assert np.array_equal(ig.ring_correlations(slabs, 128), orc.corr(128))
    finally:
        for s in slabs:
            s.close()
    cli = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")
    outs = []
    for layout in ("ballot", "dense", "nibble"):
        r = subprocess.run([cli, "-x", "8192", "-y", "64", "-n", "8", "-p", "4", "-t", "2.0", "-s", "99", "--energy", "--layout", layout],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append([ln for ln in r.stdout.splitlines() if "up_s" in ln or "energy" in ln])
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) >= 6
