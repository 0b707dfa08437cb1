"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit for bit, on the same seeded inputs.

Integer state => the bar is bit-exact: every packed word of both colours, the up/down counts and the bond sum.
"""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu

TC = ig.CRIT_TEMP_F32


def _compare(slab, orc, what):
    for color, ref in ((ig.BLACK, orc.black), (ig.WHITE, orc.white)):
        got = slab.read(color)
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            r, q = bad[0]
            raise AssertionError(f"{what}: colour {color} differs in {len(bad)} words; first at row {r} word {q}: "
                                 f"hip {int(got[r, q]):016x} oracle {int(ref[r, q]):016x}")


VARIANTS = [(ig.LAYOUT_DENSE, ig.KERNEL_AUTO), (ig.LAYOUT_DENSE, ig.KERNEL_GENERIC), (ig.LAYOUT_NIBBLE, ig.KERNEL_FAST),
            (ig.LAYOUT_NIBBLE, ig.KERNEL_GENERIC)]


@pytest.mark.parametrize("layout,kernel", VARIANTS)
@pytest.mark.parametrize("X,Y,strip", [(2048, 16, 0), (2048, 64, 4), (4096, 256, 16), (8192, 128, 0), (6144, 48, 1)])
@pytest.mark.parametrize("temp,seed", [(1.5, ig.SEED_DEF), (TC, 1234)])
def test_state_bit_exact(gpu, oracle_mod, layout, kernel, X, Y, strip, temp, seed):
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=strip, kernel=kernel, layout=layout) as s:
        assert s.layout == layout
        s.init()
        _compare(s, orc, "init")
        assert s.count() == orc.count()
        done = 0
        for upto in (1, 2, 17):
            s.sweep(upto - done)
            orc.sweep(upto - done)
            done = upto
            _compare(s, orc, f"after {upto} sweeps (kernel {kernel}, strip {s.strip_rows})")
            assert s.count() == orc.count()
            assert s.bond_equal() == orc.bond_equal()


def test_tables_match_oracle(gpu, oracle_mod):
    for temp in (1.5, 2.0, 2.26918, TC, 3.0, 10.0, 0.05 * TC):
        with ig.IsingSlab(2048, 16, temp=temp) as s:
            tab, thr = s.tables()
            assert np.array_equal(tab.view(np.uint32), oracle_mod.exp_table(temp).view(np.uint32)), temp
            # thresholds are prefixes of the draw range accepted by the FP32 test
            for a in (3, 4):
                n = thr[a]
                if 0 < n < 2**32:
                    assert oracle_mod.uniform(n - 1) <= tab[1, a] < oracle_mod.uniform(n)


@pytest.mark.parametrize("layout", [ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE])
@pytest.mark.parametrize("temp", [0.0, -1.0, 1e9, 1e-30])
def test_degenerate_temperatures_fall_back_to_generic(gpu, oracle_mod, temp, layout):
    """temp <= 0 uses the reference's special table (optimized/main.cu:1689-1693); huge temp saturates the table at 1."""
    orc = oracle_mod.OracleLattice(2048, 32, seed=7, temp=temp).init().sweep(3)
    with ig.IsingSlab(2048, 32, seed=7, temp=temp, layout=layout) as s:
        s.init().sweep(3)
        _compare(s, orc, f"temp {temp}")


def test_8192_square_config5_prefix(gpu, oracle_mod):
    """BASELINE config 5 geometry (8192^2, seed 1234), first sweeps at two temperatures: full state + series."""
    for temp in (1.5, 3.0):
        orc = oracle_mod.OracleLattice(8192, 8192, seed=1234, temp=temp).init()
        with ig.IsingSlab(8192, 8192, seed=1234, temp=temp) as s:
            s.init()
            assert s.count() == orc.count()
            for _ in range(2):
                s.sweep(2)
                orc.sweep(2)
                assert s.count() == orc.count()
                assert s.bond_equal() == orc.bond_equal()
            _compare(s, orc, f"8192^2 T={temp} after 4 sweeps")


def test_temperature_ramp(gpu, oracle_mod):
    """-u step,freq: the table is recomputed between sweeps (optimized/main.cu:1848-1859)."""
    orc = oracle_mod.OracleLattice(2048, 64, seed=99, temp=1.0).init()
    with ig.IsingSlab(2048, 64, seed=99, temp=1.0) as s:
        s.init()
        t = np.float32(1.0)
        for _ in range(6):
            s.sweep(2)
            orc.sweep(2)
            t = np.float32(max(np.float32(0.05) * np.float32(TC), t + np.float32(0.25)))
            s.set_temperature(float(t))
            orc.temp = float(t)
        _compare(s, orc, "ramp")


def test_row_range_partition_is_invariant(gpu, oracle_mod):
    """ising_update_color over arbitrary row ranges (not multiples of the strip height) + ising_update_edges must
    equal one full launch: rows of one colour are independent (each launch only reads the other colour)."""
    X, Y, seed, temp = 4096, 80, 31337, 2.1
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init().sweep(3)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, strip_rows=16) as s:
        s.init()
        for it in (1, 2, 3):
            for color in (ig.BLACK, ig.WHITE):
                s.update_edges(it, color)
                for lo, hi in ((1, 7), (7, 8), (8, 45), (45, 79)):
                    s.update_color(it, color, lo, hi)
        s.it = 3
        _compare(s, orc, "row-range partition")


@pytest.mark.parametrize("layout", [ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE])
@pytest.mark.parametrize("X,Y,XSL,YSL,strip", [(4096, 64, 2048, 16, 0), (4096, 96, 2048, 32, 16), (6144, 48, 2048, 48, 0), (4096, 64, 4096, 16, 8)])
def test_sublattices_bit_exact(gpu, oracle_mod, X, Y, XSL, YSL, strip, layout):
    """--xsl/--ysl: every XSL x YSL block is an independent torus (loadTile wrap arguments, optimized/main.cu:413-459)."""
    orc = oracle_mod.OracleLattice(X, Y, seed=8, temp=2.0, XSL=XSL, YSL=YSL).init()
    with ig.IsingSlab(X, Y, seed=8, temp=2.0, XSL=XSL, YSL=YSL, strip_rows=strip, layout=layout) as s:
        s.init()
        for n in (1, 4):
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"sub-lattices {XSL}x{YSL} after {s.it}")
            assert s.count() == orc.count()
            assert s.bond_equal() == orc.bond_equal()


def test_readme_sublattice_transcript_65536(gpu):
    """optimized/README.md:148-196: '-y 32768 -x 65536 -d 2 -t 1.5 --xsl 2048 --ysl 2048' (1024 replicas); counts are
    decomposition independent, so one 65536-row slab must print the same numbers."""
    with ig.IsingSlab(65536, 65536, seed=ig.SEED_DEF, temp=1.5, XSL=2048, YSL=2048) as s:
        s.init()
        assert s.count() == (2147484090, 2147483206)
        s.sweep(16)
        assert s.count() == (2147594634, 2147372662)   # README.md:188
        s.sweep(16)
        assert s.count() == (2147631783, 2147335513)   # README.md:189
        s.sweep(96)
        assert s.count() == (2147461873, 2147505423)   # README.md:195-196 (iter 128)


def test_write_packed_roundtrip_and_resume(gpu, oracle_mod):
    """ising_write_packed / ising_read_packed are the binary checkpoint: a state written into a fresh context (either
    device layout) continues exactly like the original run."""
    X, Y, seed, temp = 4096, 64, 99, 2.0
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init().sweep(3)
    for layout in (ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE):
        with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=layout) as s:
            s.write(ig.BLACK, orc.black)
            s.write(ig.WHITE, orc.white)
            assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
            s.it = 3
            s.sweep(4)
            ref = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init().sweep(7)
            _compare(s, ref, f"resume layout {layout}")


@pytest.mark.parametrize("layout", [ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE])
def test_large_iteration_index_counter_carry(gpu, oracle_mod, layout):
    """From iteration 2^27 on the 64-bit Philox block counter 16(2 it + colour) carries into its second word
    (curand_init offset arithmetic, optimized/main.cu:621); the kernels fold that word into round 1."""
    X, Y, seed, temp = 2048, 32, 55, 2.0
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=layout) as s:
        s.init()
        for it in (2**26 - 1, 2**27 - 1, 2**27, 2**27 + 3, 2**30 + 12345, 2**31 - 1):
            for color in (ig.BLACK, ig.WHITE):
                s.update_color(it, color)
                orc.update_color(it, color)
            _compare(s, orc, f"it={it}")


def test_wide_lattice_and_empty_ranges(gpu, oracle_mod):
    """Widest row tested (X = 262144: 128 column groups) on the minimum height, plus empty row ranges (no-ops)."""
    X, Y = 262144, 16
    orc = oracle_mod.OracleLattice(X, Y, seed=3, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=3, temp=TC) as s:
        s.init()
        assert s.count() == orc.count()
        s.update_color(1, ig.BLACK, 5, 5)   # empty range
        s.update_color(1, ig.BLACK, 0, 0)
        _compare(s, orc, "after empty launches")
        s.sweep(2)
        orc.sweep(2)
        _compare(s, orc, "wide lattice")
        assert s.bond_equal() == orc.bond_equal()


@pytest.mark.parametrize("layout", [ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE], ids=["dense", "nibble"])
@pytest.mark.parametrize("X,Y", [(2048, 256), (4096, 128), (6144, 64)])
def test_long_calls_with_temperature_changes_on_small_lattices(gpu, oracle_mod, X, Y, layout):
    """Small lattices (one launch per colour): calls of many sweeps, a temperature change between them, a temperature that needs the
    generic kernel, a private stream -- the full state equals the oracle's after every call."""
    seed, calls = 97, ((37, 2.0), (5, 2.0), (64, 1.7), (33, 0.0), (48, 2.4))
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=2.0).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=2.0, layout=layout) as s:
        s.init()
        for k, (n, t) in enumerate(calls):
            if k == 3:
                s.use_private_stream()
            s.set_temperature(t)
            s.sweep(n)
            orc.temp = float(np.float32(t))
            orc.sweep(n)
            _compare(s, orc, f"after call {k} ({n} sweeps at T = {t})")
        assert (s.count(), s.bond_equal()) == (orc.count(), orc.bond_equal())
