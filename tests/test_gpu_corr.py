"""Two-point correlations (getCorr2D_k, optimized/main.cu:870-965): exact integer sums vs the oracle; CLI file format.
The reference publishes no -c output, so the oracle's orc_corr (a literal restatement of the kernel) is UNPINNED: the
tests named *_vs_unpinned_oracle compare two restatements of the same source, not a reference vector."""
import os
import subprocess

import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")


@pytest.mark.parametrize("X,Y,temp,sweeps", [(2048, 128, 1.5, 12), (4096, 160, ig.CRIT_TEMP_F32, 5)])
def test_correlation_sums_vs_unpinned_oracle(gpu, oracle_mod, X, Y, temp, sweeps):
    orc = oracle_mod.OracleLattice(X, Y, seed=21, temp=temp).init().sweep(sweeps)
    with ig.IsingSlab(X, Y, seed=21, temp=temp) as s:
        s.init().sweep(sweeps)
        assert s.correlations(128) == orc.corr(128)
        assert s.correlations(5) == orc.corr(5)


def test_ring_correlations_match_single_slab(gpu):
    X, Y, n = 4096, 512, 4
    with ig.IsingSlab(X, Y, seed=5, temp=2.0) as one:
        one.init().sweep(3)
        ref = one.correlations(128)
    slabs = [ig.IsingSlab(X, Y // n, seed=5, temp=2.0, nslabs=n, slab=k) for k in range(n)]
    try:
        ring = ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init()
        ring.sweep(3)
        assert ig.ring_correlations(slabs, 128) == ref
    finally:
        for s in slabs:
            s.close()


def test_cli_corr_file_vs_unpinned_oracle(gpu, oracle_mod, tmp_path):
    X, Y, seed = 2048, 128, 77
    r = subprocess.run([CLI, "-x", str(X), "-y", str(Y), "-n", "8", "-p", "4", "-t", "2.0", "-s", str(seed), "-c"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr
    f = tmp_path / f"corr_{Y}x{X}_T_{2.0:f}_{seed}"
    assert f.exists(), os.listdir(tmp_path)
    lines = f.read_text().splitlines()
    assert len(lines) == 2
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=2.0).init()
    for it, line in zip((4, 8), lines):
        orc.sweep(it - orc.it)
        want = "%10d" % it + "".join(" % -12G" % (v / (2.0 * X * Y)) for v in orc.corr(128))
        assert line == want


def test_correlations_with_sublattices_vs_unpinned_oracle(gpu, oracle_mod):
    """getCorr2DRepl_k (optimized/main.cu:967-1070): wraps stay inside each XSL x YSL replica."""
    X, Y, XSL, YSL = 4096, 256, 2048, 128
    orc = oracle_mod.OracleLattice(X, Y, seed=13, temp=1.9, XSL=XSL, YSL=YSL).init().sweep(6)
    with ig.IsingSlab(X, Y, seed=13, temp=1.9, XSL=XSL, YSL=YSL) as s:
        s.init().sweep(6)
        assert s.correlations(128) == orc.corr(128)
