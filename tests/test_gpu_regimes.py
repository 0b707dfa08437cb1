"""GPU parity at the lattice shapes whose launch form layout AUTO changed at the end of round 4 -- fused launches of one-row units below 1.5 * 2^24 spins, wide-and-short
lattices on two / four ticket counters, partly dead wave columns up to a quarter, tile launches of small lattices -- against the CPU oracle: every word of both colours,
counts and bond sum after 1, 9 and 41 sweeps (calls of uneven lengths)."""
import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu

TC = ig.CRIT_TEMP_F32
B, D = ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE

CASES = [  # X, Y, layout AUTO picks, fused, tiled, strip rows (0: whatever)
    (8192, 1280, B, True, False, 1), (8192, 2048, B, True, False, 1), (16384, 768, B, True, False, 1), (16384, 2048, B, True, False, 1),
    (24576, 896, B, True, False, 1), (32768, 640, B, True, False, 1), (32768, 1024, B, True, False, 1), (65536, 512, B, True, False, 1),
    (16384, 2176, B, True, False, 2), (24576, 1536, B, True, False, 2), (131072, 1024, B, True, False, 4), (24576, 4096, B, True, False, 2),
    (12288, 1536, B, True, False, 1), (6144, 3072, B, True, False, 1), (20480, 4096, B, True, False, 0), (28672, 4096, B, True, False, 0),
    (8192, 1024, D, False, True, 0), (4096, 4096, D, False, True, 0), (6144, 1024, D, False, True, 0), (2048, 8192, D, False, True, 0), (65536, 256, D, False, True, 0),
]


@pytest.mark.parametrize("X,Y,layout,fused,tiled,H", CASES)
def test_auto_regimes_bit_exact(gpu, oracle_mod, X, Y, layout, fused, tiled, H):
    oracle_mod.set_threads(16)
    orc = oracle_mod.OracleLattice(X, Y, seed=4242, temp=TC).init()
    with ig.IsingSlab(X, Y, seed=4242, temp=TC) as s:
        assert (s.layout, s.fused, s.tiled) == (layout, fused, tiled), (s.layout, s.fused, s.tiled, s.strip_rows)
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
