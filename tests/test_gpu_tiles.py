"""GPU parity of the tile launches (several sweeps per launch for lattices on the dense layout, ising_dense.hip: dense_tile_k) against the
CPU oracle, bit for bit: every word of both colours, counts and bond sum -- over tile shapes, sweeps per launch (halo depth), workgroup
sizes, calls that split unevenly into launches, lattices a tile's halo wraps around, and the counter's high word."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu

TC = ig.CRIT_TEMP_F32
KEYS = ("ISING_TILES", "ISING_TILE_ROWS", "ISING_TILE_WORDS", "ISING_TILE_SWEEPS", "ISING_TILE_THREADS", "ISING_TILE_XCD")


def _env(monkeypatch, **kw):
    for k in KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ISING_QUAD", "0")  # (round 5: lattices up to four blocks of 2048 columns take the quad path by default -- tests/test_gpu_quad.py; these are the tile launches')
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


SHAPES = [  # X, Y, tile rows, tile words, sweeps per launch, threads
    (2048, 64, 8, 16, 2, 256), (2048, 64, 16, 32, 4, 512), (2048, 16, 16, 16, 8, 256), (2048, 32, 8, 32, 16, 1024),
    (4096, 128, 32, 32, 4, 1024), (4096, 64, 16, 64, 3, 512), (6144, 48, 8, 32, 5, 256), (8192, 64, 64, 32, 6, 1024), (4096, 256, 8, 16, 7, 256),
]


@pytest.mark.parametrize("X,Y,TR,TWI,S,NT", SHAPES)
@pytest.mark.parametrize("temp,seed", [(1.5, ig.SEED_DEF), (TC, 1234)])
def test_tiles_bit_exact(gpu, oracle_mod, monkeypatch, X, Y, TR, TWI, S, NT, temp, seed):
    _env(monkeypatch, TILES=1, TILE_ROWS=TR, TILE_WORDS=TWI, TILE_SWEEPS=S, TILE_THREADS=NT)
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_DENSE) as s:
        s.init()
        done = 0
        for upto in (2, 3, 3 + 2 * S, 4 + 4 * S + 1, 40):  # two launches of one sweep; one per colour; exactly two full launches; uneven; many
            s.sweep(upto - done)
            orc.sweep(upto - done)
            done = upto
            _compare(s, orc, f"after {upto} sweeps ({TR} x {TWI} tiles, {S} sweeps a launch, {NT} threads)")


def test_tiles_default_rule(gpu, oracle_mod, monkeypatch):
    """Lone slabs on the dense layout up to 2^24 spins sweep in tile launches by default (ising_sweep_info: 2); ISING_TILES=0, larger
    lattices, ring slabs, couplings, sub-lattices and temperatures without integer thresholds keep one launch per colour."""
    _env(monkeypatch)
    orc = oracle_mod.OracleLattice(2048, 512, seed=99, temp=TC).init()
    orc.sweep(23)
    for env, tiled in ((dict(), True), (dict(TILE_ROWS=8, TILE_XCD=0), True), (dict(TILES=0), False)):
        _env(monkeypatch, **env)
        with ig.IsingSlab(2048, 512, seed=99, temp=TC) as s:
            assert s.layout == ig.LAYOUT_DENSE and s.tiled == tiled and not s.fused
            assert s.max_sweeps_per_launch == (3 if tiled else 0)  # (8-row tiles: 3 sweeps a launch)
            s.init().sweep(23)
            _compare(s, orc, str(env))
    _env(monkeypatch)
    with ig.IsingSlab(4096, 4096, temp=TC) as s:
        assert s.tiled and s.max_sweeps_per_launch == 6  # (64-row tiles)
    with ig.IsingSlab(4096, 4096 + 2048, temp=TC) as s:  # 1.5 * 2^24 spins, dense layout: above the rule
        assert s.layout == ig.LAYOUT_DENSE and not s.tiled
    with ig.IsingSlab(2048, 512, temp=TC, nslabs=2, slab=0) as s:
        assert not s.tiled
    with ig.IsingSlab(2048, 512, temp=TC, J_prob=0.1) as s:
        assert not s.tiled
    with ig.IsingSlab(4096, 512, temp=TC, XSL=2048, YSL=256) as s:
        assert not s.tiled
    with ig.IsingSlab(2048, 512, temp=TC, kernel=ig.KERNEL_GENERIC) as s:
        assert not s.tiled
    with ig.IsingSlab(2048, 512, temp=0.0) as s:
        assert not s.tiled


@pytest.mark.parametrize("it0", [(1 << 27) - 3, (1 << 30) + 12345, (1 << 31) - 12])
def test_tiles_counter_high_word(gpu, oracle_mod, monkeypatch, it0):
    """Iterations from 2^27 on: the draw-block counter 16 (2 it + colour) needs its high word (optimized/main.cu:621: the offset is 64 bits)."""
    _env(monkeypatch, TILES=1, TILE_SWEEPS=4)
    orc = oracle_mod.OracleLattice(2048, 64, seed=5, temp=TC).init()
    with ig.IsingSlab(2048, 64, seed=5, temp=TC, layout=ig.LAYOUT_DENSE) as s:
        s.init()
        s.it = orc.it = it0
        s.sweep(9)
        orc.sweep(9)
        _compare(s, orc, "across iteration 2^27")


@pytest.mark.parametrize("X,Y,first,n,every", [(2048, 64, 0, 40, 16), (2048, 128, 5, 37, 7), (4096, 64, 0, 9, 1), (2048, 32, 3, 50, 100), (2048, 16, 0, 33, 16)])
def test_tiles_counted_sweeps(gpu, oracle_mod, monkeypatch, X, Y, first, n, every):
    """ising_sweep_counted on a tiled slab: the print points are ends of tile launches whose workgroups count what they store -- the
    counts of every iteration that is a multiple of `every`, and the state afterwards (odd numbers of launches: the copy back)."""
    _env(monkeypatch)
    orc = oracle_mod.OracleLattice(X, Y, seed=77, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=77, temp=TC) as s:
        assert s.tiled
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


def test_tiles_randomised(gpu, oracle_mod, monkeypatch):
    """Random lattices, tile shapes, halo depths, workgroup sizes and call lengths against the oracle (seeded: the same 48 cases every run)."""
    rng = np.random.default_rng(20260929)
    for case in range(48):
        gx = int(rng.integers(1, 4))
        X = 2048 * gx
        Y = 16 * int(rng.integers(1, 13))
        TR = int(rng.choice([t for t in (8, 16, 24, 32, 48, 64, 96, 128) if Y % t == 0]))
        TWI = int(rng.choice([w for w in (8, 12, 16, 24, 32, 48, 64, 96) if (32 * gx) % w == 0]))
        S = int(rng.integers(1, min(16, Y // 2) + 1))
        lds = lambda S: (2 * (TR + 4 * S) * (TWI + 2) + 2 * S * 16 * 3) * 4  # noqa: E731  (both colours of tile + halo, the block constants)
        while S > 1 and lds(S) > 65536:
            S -= 1
        if lds(S) > 65536:
            continue
        NT = int(rng.choice([256, 512, 1024]))
        temp = float(rng.choice([1.5, 2.0, TC, 3.0]))
        seed = int(rng.integers(1, 2**62))
        _env(monkeypatch, TILES=1, TILE_ROWS=TR, TILE_WORDS=TWI, TILE_SWEEPS=S, TILE_THREADS=NT, TILE_XCD=int(rng.integers(0, 2)))
        orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=temp).init()
        with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=ig.LAYOUT_DENSE) as s:
            assert s.tiled and s.max_sweeps_per_launch == S
            s.init()
            for n in rng.integers(1, 3 * S + 4, size=3):
                if rng.integers(0, 3) == 0:
                    every = int(rng.integers(1, 9))
                    got = s.sweep_counted(int(n), every)
                    want = []
                    for _ in range(int(n)):
                        orc.sweep(1)
                        if orc.it % every == 0:
                            want.append(orc.count())
                    assert got == want, (case, X, Y, TR, TWI, S, NT)
                else:
                    s.sweep(int(n))
                    orc.sweep(int(n))
                _compare(s, orc, f"case {case}: {Y} x {X}, tiles {TR} x {TWI}, {S} sweeps a launch, {NT} threads, after {orc.it} sweeps")


@pytest.mark.parametrize("temp", [0.0, -1.0])
def test_tiles_fast_kernel_without_thresholds_is_an_error(gpu, monkeypatch, temp):
    """ISING_KERNEL_FAST at a temperature whose table has no integer thresholds (T <= 0): the header promises an error, and the tile
    launches -- which know integer thresholds only -- must not run on truncated ones (round-4 ADVICE)."""
    _env(monkeypatch)
    with ig.IsingSlab(2048, 2048, seed=7, temp=temp, layout=ig.LAYOUT_DENSE, kernel=ig.KERNEL_FAST) as s:
        s.init()
        before = s.read(ig.BLACK).copy()
        with pytest.raises(ig.IsingError):
            s.sweep(4)
        with pytest.raises(ig.IsingError):
            s.sweep_counted(4, 2)
        with pytest.raises(ig.IsingError):
            s.sweep_counted(4, 2, True)
        s.synchronize()
        assert np.array_equal(s.read(ig.BLACK), before), "a refused sweep changed the lattice"
