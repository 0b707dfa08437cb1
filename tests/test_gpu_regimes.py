"""GPU parity at the lattice shapes whose launch form layout AUTO changed at the end of round 4 -- fused launches of one-row units below 1.5 * 2^24 spins, wide-and-short
lattices on two / four ticket counters, partly dead wave columns up to a quarter, tile launches of small lattices -- against the CPU oracle: every word of both colours,
counts and bond sum after 1, 9 and 41 sweeps (calls of uneven lengths)."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu

TC = ig.CRIT_TEMP_F32
B, D = ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE

CASES = [  # X, Y, layout AUTO picks, fused, tiled, strip rows (0: whatever); quad (round 5: up to four blocks of 2048 columns, ising_quad.hip) = dense, neither fused nor tiled
    (8192, 1280, B, True, False, 1), (8192, 2048, B, True, False, 1), (16384, 768, B, True, False, 1), (16384, 2048, B, True, False, 1),
    (24576, 896, B, True, False, 1), (32768, 640, B, True, False, 1), (32768, 1024, B, True, False, 1), (65536, 512, B, True, False, 1),
    (16384, 2176, B, True, False, 2), (24576, 1536, B, True, False, 2), (131072, 1024, B, True, False, 4), (24576, 4096, B, True, False, 4),
    (12288, 1536, B, True, False, 1), (6144, 3072, D, False, False, 0), (20480, 4096, B, True, False, 0), (28672, 4096, B, True, False, 0),
    (8192, 1024, D, False, False, 0), (4096, 4096, D, False, False, 0), (6144, 1024, D, False, False, 0), (2048, 8192, D, False, False, 0), (65536, 256, D, False, True, 0),
    (8192, 4096, B, True, False, 2), (10240, 1024, D, False, False, 0),
]


@pytest.mark.parametrize("X,Y,layout,fused,tiled,H", CASES)
def test_auto_regimes_bit_exact(gpu, oracle_mod, X, Y, layout, fused, tiled, H):
    oracle_mod.set_threads(16)
    orc = oracle_mod.OracleLattice(X, Y, seed=4242, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=4242, temp=TC) as s:
        assert (s.layout, s.fused, s.tiled) == (layout, fused, tiled), (s.layout, s.fused, s.tiled, s.strip_rows)
        assert s.quad == (layout == D and not tiled and X <= 16384), s.quad
        assert not H or s.strip_rows == H, s.strip_rows
        s.init()
        done = 0
        for upto in (1, 9, 41):
            s.sweep(upto - done)
            orc.sweep(upto - done)
            done = upto
            for color, ref in ((ig.BLACK, orc.black), (ig.WHITE, orc.white)):
                got = s.read(color)
                if not np.array_equal(got, ref):
                    bad = np.argwhere(got != ref)
                    raise AssertionError(f"{Y} x {X} after {upto} sweeps: colour {color} differs in {len(bad)} words, first at {tuple(bad[0])}")
            assert s.count() == orc.count() and s.bond_equal() == orc.bond_equal()


def test_auto_random_shapes_bit_exact(gpu, oracle_mod):
    """Layout AUTO on 40 random lattice shapes up to 2^27 spins (seeded: the same every run) -- whatever layout, launch form, strip height, grid and ticket counters
    ising_create picks -- against the CPU oracle after 1, 7 and 33 sweeps; the counted sweeps' print points too."""
    oracle_mod.set_threads(16)
    rng = np.random.default_rng(4)
    seen = set()
    for case in range(40):
        while True:
            X = 2048 * int(rng.integers(1, 41))
            Y = 16 * int(rng.integers(1, 513))
            if X * Y <= (1 << 27) and X * Y >= (1 << 19):
                break
        seed = int(rng.integers(1, 2**62))
        orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=TC).init()
        with ig.IsingSlab(X, Y, seed=seed, temp=TC) as s:
            seen.add((s.layout, s.fused, s.tiled, s.quad, s.strip_rows))
            s.init()
            done = 0
            for upto in (1, 7, 33):
                if upto == 33:
                    got = s.sweep_counted(upto - done, 16)
                    want = []
                    for _ in range(upto - done):
                        orc.sweep(1)
                        if orc.it % 16 == 0:
                            want.append(orc.count())
                    assert got == want, (case, X, Y)
                else:
                    s.sweep(upto - done)
                    orc.sweep(upto - done)
                done = upto
                ok = np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white) and s.count() == orc.count() and s.bond_equal() == orc.bond_equal()
                assert ok, f"case {case}: {Y} x {X} after {upto} sweeps (layout {s.layout}, fused {s.fused}, tiled {s.tiled}, H = {s.strip_rows})"
    assert len(seen) >= 4, seen  # (the cases do spread over the launch forms)
