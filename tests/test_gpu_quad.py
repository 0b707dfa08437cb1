"""GPU parity of the quad path (small lattices: one launch per pass of several sweeps = the word pass on tiles + halo, next to the draws of the pass to
come; ising_quad.hip, ising_update.cpp: sweep_quad) against the CPU oracle, bit for bit: every word of both colours, counts and bond sum -- over lattice
widths (1 .. 4 blocks of 2048 columns), tile heights, sweeps per pass (halo depth), waves per workgroup, calls that
split unevenly into passes, lattices a tile's halo wraps around (several times), and the counter's high word."""
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


def _compare(slab, orc, what):
    for color, ref in ((ig.BLACK, orc.black), (ig.WHITE, orc.white)):
        got = slab.read(color)
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            r, q = bad[0]
            raise AssertionError(f"{what}: colour {color} differs in {len(bad)} words; first at row {r} word {q}: hip {int(got[r, q]):016x} oracle {int(ref[r, q]):016x}")
    assert slab.count() == orc.count(), what
    assert slab.bond_equal() == orc.bond_equal(), what


SHAPES = [  # X, Y, row groups per tile, sweeps per pass, waves
    (2048, 64, 8, 8, 4), (2048, 64, 4, 4, 8), (2048, 16, 4, 2, 2), (2048, 32, 2, 16, 16), (2048, 128, 16, 3, 8),
    (4096, 128, 8, 8, 8), (4096, 64, 8, 5, 8), (6144, 48, 3, 6, 16), (8192, 64, 4, 4, 8), (4096, 256, 7, 7, 16), (4096, 4096, 4, 8, 12), (2048, 8192, 4, 8, 8),
    (10240, 64, 2, 2, 12), (12288, 128, 2, 2, 12), (14336, 64, 2, 2, 16), (16384, 48, 2, 2, 16),
    (6144, 128, 4, 8, 16), (8192, 256, 2, 8, 16), (12288, 64, 2, 6, 16), (10240, 128, 4, 4, 16), (16384, 64, 2, 4, 16), (6144, 256, 8, 6, 16),
]


@pytest.mark.parametrize("X,Y,C,T,NW", SHAPES)
@pytest.mark.parametrize("temp,seed", [(1.5, ig.SEED_DEF), (TC, 1234)])
def test_quad_bit_exact(gpu, oracle_mod, monkeypatch, X, Y, C, T, NW, temp, seed):
    _env(monkeypatch, QUAD=1, QUAD_C=C, QUAD_T=T, QUAD_WAVES=NW)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_DENSE) as s:
        assert s.quad and s.max_sweeps_per_launch == T
        s.init()
        done = 0
        for upto in (2, 3, 3 + 2 * T, 4 + 4 * T + 1, 4 + 4 * T + 1 + 7 * T + 1):  # one short pass; one per colour; exactly two full passes; uneven; many
            s.sweep(upto - done)
            orc.sweep(upto - done)
            done = upto
            _compare(s, orc, f"after {upto} sweeps (tiles of {C} row groups, {T} sweeps a pass, {NW} waves)")


@pytest.mark.parametrize("it0", [(1 << 27) - 3, (1 << 30) + 12345, (1 << 31) - 12])
def test_quad_counter_high_word(gpu, oracle_mod, monkeypatch, it0):
    """Iterations from 2^27 on: the draw-block counter 16 (2 it + colour) needs its high word (optimized/main.cu:621: the offset is 64 bits)."""
    _env(monkeypatch, QUAD=1, QUAD_T=4)
    orc = oracle_mod.OracleLattice(2048, 64, seed=5, temp=TC).init()
    with ig.IsingSlab(2048, 64, seed=5, temp=TC, layout=ig.LAYOUT_DENSE) as s:
        assert s.quad
        s.init()
        s.it = orc.it = it0
        s.sweep(9)
        orc.sweep(9)
        _compare(s, orc, "across iteration 2^27")


@pytest.mark.parametrize("X,Y,first,n,every", [(2048, 64, 0, 40, 16), (2048, 128, 5, 37, 7), (4096, 64, 0, 9, 1), (2048, 32, 3, 50, 100), (2048, 16, 0, 33, 16)])
def test_quad_counted_sweeps(gpu, oracle_mod, monkeypatch, X, Y, first, n, every):
    """ising_sweep_counted on the quad path: the print points are ends of word passes whose workgroups count what they store."""
    _env(monkeypatch, QUAD=1)
    orc = oracle_mod.OracleLattice(X, Y, seed=77, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=77, temp=TC, layout=ig.LAYOUT_DENSE) as s:
        assert s.quad
        s.init()
        s.sweep(first)
        orc.sweep(first)
        got = s.sweep_counted(n, every)
        want = []
        for _ in range(n):
            orc.sweep(1)
            if orc.it % every == 0:
                want.append(orc.count())
        assert got == want
        _compare(s, orc, "after the counted sweeps")
        s.sweep(3)
        orc.sweep(3)
        _compare(s, orc, "three sweeps later")


@pytest.mark.parametrize("X,Y,first,n,every", [(2048, 64, 0, 40, 16), (4096, 128, 5, 37, 7), (6144, 48, 0, 9, 1), (8192, 32, 3, 50, 25), (2048, 2048, 0, 33, 16)])
def test_quad_counted_sweeps_with_the_energy(gpu, oracle_mod, monkeypatch, X, Y, first, n, every):
    """... with the bond sum (north_star's energy series) at the same print points: the tiles of a measured pass count, behind its last level, the equal
    neighbours of their white sites (every bond has one white end) -- no fall-back to sweeping and measuring in turn."""
    _env(monkeypatch, QUAD=1)
    orc = oracle_mod.OracleLattice(X, Y, seed=78, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=78, temp=TC, layout=ig.LAYOUT_DENSE) as s:
        assert s.quad
        s.init()
        s.sweep(first)
        orc.sweep(first)
        got = s.sweep_counted(n, every, True)
        want = []
        for _ in range(n):
            orc.sweep(1)
            if orc.it % every == 0:
                want.append(orc.count() + (orc.bond_equal(),))
        assert got == want
        _compare(s, orc, "after the counted sweeps")


def test_quad_temperature_change_between_calls(gpu, oracle_mod, monkeypatch):
    """The draws of a call carry the thresholds of the temperature the call was made at (the `-u` ramp, optimized/main.cu:1848-1860)."""
    _env(monkeypatch, QUAD=1)
    orc = oracle_mod.OracleLattice(2048, 128, seed=3, temp=1.5).init()
    with ig.IsingSlab(2048, 128, seed=3, temp=1.5, layout=ig.LAYOUT_DENSE) as s:
        s.init()
        for temp, n in ((1.5, 20), (2.0, 11), (TC, 30), (3.0, 4)):
            s.set_temperature(temp)
            orc.temp = float(np.float32(temp))
            s.sweep(n)
            orc.sweep(n)
            _compare(s, orc, f"T = {temp}")


def test_quad_randomised(gpu, oracle_mod, monkeypatch):
    """Random lattices, tile heights, halo depths, workgroup sizes and call lengths against the oracle (seeded: the same 48 cases every run)."""
    rng = np.random.default_rng(20260930)
    done = 0
    for case in range(64):
        gx = int(rng.integers(1, 5))
        X = 2048 * gx
        Y = 16 * int(rng.integers(1, 13))
        T = int(rng.integers(1, 9 if gx > 1 else 13))
        C = int(rng.integers(1, 9 if gx > 1 else 17))
        NW = int(rng.choice([4, 8, 12, 16] if gx > 1 else [1, 2, 4, 8, 12, 16]))
        HG = (2 * T + 2) // 4
        items = (min(C, Y // 4) + 2 * HG) * gx
        per_wave = (items + NW - 1) // NW
        if per_wave > 4 or (per_wave > 3 and NW > 8) or items * 1024 > 150 * 1024:
            continue
        temp = float(rng.choice([1.5, 2.0, TC, 3.0]))
        seed = int(rng.integers(1, 2**62))
        _env(monkeypatch, QUAD=1, QUAD_C=C, QUAD_T=T, QUAD_WAVES=NW)
        orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
        with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_DENSE) as s:
            assert s.quad and s.max_sweeps_per_launch == T
            s.init()
            for n in rng.integers(2, 5 * T + 4, size=3):
                if rng.integers(0, 3) == 0:
                    every = int(rng.integers(1, 9))
                    energy = bool(rng.integers(0, 2))
                    got = s.sweep_counted(int(n), every, energy)
                    want = []
                    for _ in range(int(n)):
                        orc.sweep(1)
                        if orc.it % every == 0:
                            want.append(orc.count() + ((orc.bond_equal(),) if energy else ()))
                    assert got == want, (case, X, Y, C, T, NW)
                else:
                    s.sweep(int(n))
                    orc.sweep(int(n))
                _compare(s, orc, f"case {case}: {Y} x {X}, tiles of {C} row groups, {T} sweeps a pass, {NW} waves, after {orc.it} sweeps")
        done += 1
    assert done >= 20, done


def test_quad_default_rule(gpu, oracle_mod, monkeypatch):
    """Lone slabs of up to six blocks of 2048 columns sweep on the quad path by default (ising_sweep_info: 4) -- up to 2^26 spins for one and two blocks,
    6144 rows for three, 1024 rows for four to six, 512 for seven and eight; ISING_QUAD=0, wider or larger lattices, ring slabs, couplings, sub-lattices, the generic kernel, temperatures without integer
    thresholds and the other layouts asked for by name keep what they had."""
    _env(monkeypatch)
    orc = oracle_mod.OracleLattice(2048, 512, seed=99, temp=TC).init()
    orc.sweep(23)
    for env, quad in ((dict(), True), (dict(QUAD_T=3, QUAD_C=2), True), (dict(QUAD=0), False)):
        _env(monkeypatch, **env)
        with ig.IsingSlab(2048, 512, seed=99, temp=TC) as s:
            assert s.layout == ig.LAYOUT_DENSE and s.quad == quad and not s.fused and s.tiled == (not quad)
            s.init().sweep(23)
            _compare(s, orc, str(env))
    _env(monkeypatch)
    for X, Y, quad in ((4096, 16384, True), (6144, 6144, True), (6144, 8192, False), (8192, 1024, True), (8192, 2048, False), (10240, 1024, True), (12288, 512, True), (12288, 2048, False), (14336, 1024, False), (14336, 512, True), (16384, 512, True), (16384, 768, False), (6144, 1024, True), (6144, 2048, True), (8192, 512, True), (4096, 32768, False), (2048, 16, True)):
        with ig.IsingSlab(X, Y, temp=TC) as s:
            assert s.quad == quad, (X, Y)
            sweeps_a_pass = {2048: 8 if Y < 1024 else (16 if Y < 4096 else (12 if Y < 8192 else (8 if Y < 16384 else 4))), 4096: 12 if Y < 2048 else (8 if Y < 8192 else 4),
                             6144: 8 if Y < 2048 else (6 if Y < 4096 else 4), 8192: 8, 10240: 6 if Y < 1024 else 4, 12288: 6 if Y < 1024 else 4, 14336: 4, 16384: 4}.get(X)
            assert not quad or (s.layout == ig.LAYOUT_DENSE and s.max_sweeps_per_launch == sweeps_a_pass)
    with ig.IsingSlab(2048, 512, temp=TC, layout=ig.LAYOUT_DENSE) as s:
        assert s.quad
    for kw in (dict(nslabs=2, slab=0), dict(J_prob=0.1), dict(kernel=ig.KERNEL_GENERIC), dict(layout=ig.LAYOUT_NIBBLE), dict(layout=ig.LAYOUT_BALLOT)):
        with ig.IsingSlab(8192, 512, temp=TC, **kw) as s:
            assert not s.quad, kw
    with ig.IsingSlab(4096, 512, temp=TC, XSL=2048, YSL=256) as s:
        assert not s.quad
    with ig.IsingSlab(2048, 512, temp=0.0) as s:
        assert not s.quad
