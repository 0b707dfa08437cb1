"""-J (random anti-ferromagnetic bonds): coupling arrays and the coupled update against the oracle's literal
restatement of hamiltInitB_k / hamiltInitW_k / the jDst branch of spinUpdateV_2D_k (optimized/main.cu:153-331, :575-618).
The reference has no transcript for -J, so these vectors come from the oracle only."""
import os
import re
import subprocess

import numpy as np
import pytest

import ising_gpu_amd as ig

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(ig.LIB_PATH), "cuIsing")
# the coupling arrays are nibble-packed on the device with the reference's layout and four bit-planes with the dense one
LAYOUTS = pytest.mark.parametrize("layout", [ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE], ids=["dense", "nibble"])


@pytest.mark.parametrize("X,Y,prob,kernel", [(2048, 32, 0.3, ig.KERNEL_AUTO), (4096, 64, 0.5, ig.KERNEL_GENERIC), (6144, 48, 1.0, ig.KERNEL_AUTO),
                                             (2048, 16, 0.0, ig.KERNEL_AUTO)])
@LAYOUTS
def test_couplings_and_update_vs_unpinned_oracle(gpu, oracle_mod, X, Y, prob, kernel, layout):
    orc = oracle_mod.OracleLattice(X, Y, seed=1234, temp=1.8).init().init_couplings(prob)
    with ig.IsingSlab(X, Y, seed=1234, temp=1.8, J_prob=prob, kernel=kernel, layout=layout) as s:
        assert s.layout == layout
        s.init().init_couplings()
        assert np.array_equal(s.read_couplings(ig.BLACK), orc.hamB)
        assert np.array_equal(s.read_couplings(ig.WHITE), orc.hamW)
        for n in (1, 5):
            s.sweep(n)
            orc.sweep(n)
            assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white), (prob, s.it)


ALL3 = pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE], ids=["ballot", "dense", "nibble"])


@ALL3
@pytest.mark.parametrize("X,Y,prob,sub", [(8192, 64, 0.3, None), (16384, 32, 0.7, None), (8192, 64, 0.45, (4096, 32))])
def test_couplings_vs_the_second_restatement(gpu, X, Y, prob, sub, layout):
    """ising_init_couplings (ham_init_black_k / ham_init_white_k and the layout's plane transposition, read back through
    ising_read_couplings) against tests/_couplings_np.py -- the closed-form restatement that never touched the oracle's code
    (VERDICT r03 item 6): seed + 1, draw order nibble -> bit -> word x, y, `u < prob`; white bits = the black neighbours' bits of the
    same bonds."""
    from _couplings_np import ham_black_np, ham_white_np
    seed = 20240911
    kw = dict(XSL=sub[0], YSL=sub[1]) if sub else {}
    hb = ham_black_np(X, Y, 0, seed + 1, float(np.float32(prob)))
    hw = ham_white_np(hb, *(sub or (0, 0)))
    with ig.IsingSlab(X, Y, seed=seed, temp=1.5, J_prob=prob, layout=layout, **kw) as s:
        assert s.layout == layout
        s.init().init_couplings()
        assert np.array_equal(s.read_couplings(ig.BLACK), hb)
        assert np.array_equal(s.read_couplings(ig.WHITE), hw)


def test_ring_couplings_vs_the_second_restatement(gpu):
    """The same on a ring of three slabs with ghost rows (every slab generates its rows -- and its ghost rows' -- at their global row)."""
    from _couplings_np import ham_black_np, ham_white_np
    X, Y, n, seed, prob = 8192, 384, 3, 77, 0.35
    hb = ham_black_np(X, Y, 0, seed + 1, float(np.float32(prob)))
    hw = ham_white_np(hb)
    slabs = [ig.IsingSlab(X, Y // n, seed=seed, temp=1.5, nslabs=n, slab=k, J_prob=prob, layout=ig.LAYOUT_BALLOT) for k in range(n)]
    try:
        ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init()
        assert np.array_equal(np.concatenate([s.read_couplings(ig.BLACK) for s in slabs]), hb)
        assert np.array_equal(np.concatenate([s.read_couplings(ig.WHITE) for s in slabs]), hw)
    finally:
        for s in slabs:
            s.close()


@LAYOUTS
def test_couplings_with_sublattices_vs_unpinned_oracle(gpu, oracle_mod, layout):
    X, Y = 4096, 64
    orc = oracle_mod.OracleLattice(X, Y, seed=9, temp=1.2, XSL=2048, YSL=32).init().init_couplings(0.4)
    with ig.IsingSlab(X, Y, seed=9, temp=1.2, XSL=2048, YSL=32, J_prob=0.4, layout=layout) as s:
        s.init().init_couplings()
        assert np.array_equal(s.read_couplings(ig.WHITE), orc.hamW)
        s.sweep(3)
        orc.sweep(3)
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)


@LAYOUTS
def test_couplings_ring_matches_single_slab(gpu, layout):
    X, Y, n = 4096, 192, 3
    with ig.IsingSlab(X, Y, seed=4, temp=1.5, J_prob=0.35, layout=layout) as one:
        one.init().init_couplings().sweep(4)
        ref = (one.read(ig.BLACK), one.read(ig.WHITE), one.read_couplings(ig.WHITE))
    slabs = [ig.IsingSlab(X, Y // n, seed=4, temp=1.5, nslabs=n, slab=k, J_prob=0.35, layout=layout) for k in range(n)]
    try:
        ring = ig.LocalRing([ig.HipSlabBackend(s) for s in slabs]).init()
        ring.sweep(4)
        assert np.array_equal(np.concatenate([s.read_couplings(ig.WHITE) for s in slabs]), ref[2])
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in slabs]), ref[0])
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in slabs]), ref[1])
    finally:
        for s in slabs:
            s.close()
    # the same with torch-owned spin and coupling buffers (what SlabRing hands to RCCL)
    backs = [ig.HipSlabBackend.create(X, Y // n, seed=4, temp=1.5, nslabs=n, slab=k, J_prob=0.35, layout=layout) for k in range(n)]
    try:
        ig.LocalRing(backs).init().sweep(4)
        assert np.array_equal(np.concatenate([b.slab.read_couplings(ig.WHITE) for b in backs]), ref[2])
        assert np.array_equal(np.concatenate([b.slab.read(ig.WHITE) for b in backs]), ref[1])
    finally:
        for b in backs:
            b.slab.close()


def test_cli_J_transcript_vs_unpinned_oracle(gpu, oracle_mod):
    X, Y, seed = 2048, 64, 606
    for ndev, ylocal, extra in ((1, Y, []), (1, Y, ["--layout", "nibble"]), (2, Y // 2, ["--devmap", "0,0"])):
        r = subprocess.run([CLI, "-x", str(X), "-y", str(ylocal), "-d", str(ndev), "-n", "6", "-p", "3", "-t", "1.0", "-s", str(seed), "-J", "0.25"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "\tusing Hamiltonian buffer, setting links to -1 with prob 0.25\n" in r.stdout
        assert "not using Hamiltonian buffer" not in r.stdout  # exactly one of the two lines, optimized/main.cu:1577-1581
        orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=1.0).init().init_couplings(0.25)
        for it in (3, 6):
            orc.sweep(it - orc.it)
            up, dw = orc.count()
            assert f"up_s: {up:12d}, dw_s: {dw:12d} (iter: {it:8d})" in r.stdout, (ndev, it)


ALL_LAYOUTS = pytest.mark.parametrize("layout", [ig.LAYOUT_BALLOT, ig.LAYOUT_DENSE, ig.LAYOUT_NIBBLE], ids=["ballot", "dense", "nibble"])


@ALL_LAYOUTS
def test_written_couplings_read_back_and_drive_the_update(gpu, oracle_mod, layout):
    """ising_write_couplings: arrays in the reference's nibble form go to the device form of each layout and come back unchanged;
    the update that follows uses them (against the oracle given the same arrays).  One array written into a fresh context, then
    the other; then one array replaced."""
    X, Y, seed = 8192, 64, 606
    rng = np.random.default_rng(5)
    hb, hw, hb2 = (rng.integers(0, 2**63, size=(Y, X // 32), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(Y, X // 32), dtype=np.uint64)
                   for _ in range(3))
    orc = oracle_mod.OracleLattice(X, Y, seed=seed, temp=1.7).init().init_couplings(0.0)
    with ig.IsingSlab(X, Y, seed=seed, temp=1.7, J_prob=0.5, layout=layout) as s:
        s.init()
        s.write_couplings(ig.BLACK, hb)
        assert np.array_equal(s.read_couplings(ig.BLACK), hb) and not s.read_couplings(ig.WHITE).any()
        s.write_couplings(ig.WHITE, hw)
        assert np.array_equal(s.read_couplings(ig.BLACK), hb) and np.array_equal(s.read_couplings(ig.WHITE), hw)
        orc.hamB[:], orc.hamW[:] = hb, hw
        s.sweep(3)
        orc.sweep(3)
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
        s.write_couplings(ig.BLACK, hb2)
        assert np.array_equal(s.read_couplings(ig.BLACK), hb2) and np.array_equal(s.read_couplings(ig.WHITE), hw)
        orc.hamB[:] = hb2
        s.sweep(2)
        orc.sweep(2)
        assert np.array_equal(s.read(ig.BLACK), orc.black) and np.array_equal(s.read(ig.WHITE), orc.white)
    with ig.IsingSlab(X, Y, seed=seed, temp=1.7, layout=layout) as s, pytest.raises(ig.IsingError):
        s.write_couplings(ig.BLACK, hb)  # couplings not enabled


@ALL_LAYOUTS
def test_coupled_update_against_the_plain_one_by_exact_maps(gpu, layout):
    """No reference vector exists for -J, so the HIP coupling path is tied to the pinned plain path at a size the oracle does not
    reach, by two exact maps (see tests/test_oracle_kat.py): all coupling bits clear -> the plain trajectory; all bits set (every
    bond antiferromagnetic) and the black colour complemented -> the plain trajectory with the black colour complemented."""
    X, Y, seed, temp, n = 16384, 2048, 99, 2.1, 6
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=layout) as s:
        s.init().sweep(n)
        pb, pw = s.read(ig.BLACK), s.read(ig.WHITE)
    ones = np.uint64(0x1111111111111111)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, J_prob=0.0, layout=layout) as s:
        s.init().init_couplings()
        s.sweep(n)
        assert np.array_equal(s.read(ig.BLACK), pb) and np.array_equal(s.read(ig.WHITE), pw)
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, J_prob=1.0, layout=layout) as s:
        s.init()
        full = np.full((Y, X // 32), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        s.write_couplings(ig.BLACK, full)
        s.write_couplings(ig.WHITE, full)
        s.write(ig.BLACK, s.read(ig.BLACK) ^ ones)
        s.sweep(n)
        assert np.array_equal(s.read(ig.BLACK) ^ ones, pb) and np.array_equal(s.read(ig.WHITE), pw)


@ALL_LAYOUTS
def test_coupled_update_under_a_random_gauge(gpu, layout):
    """tests/test_oracle_kat.py::test_coupling_path_under_a_random_gauge on the GPU, 16384 x 2048 on every device layout: Mattis
    couplings of a random gauge field (each colour's update given its own sites' bond nibbles: black sites' into the array the
    black update reads, ISING_WHITE) and the gauge-transformed start reproduce the gauge-transformed plain trajectory."""
    from _gauge import gauge_field, gauge_words, mattis_nibbles
    X, Y, seed, temp, n = 16384, 2048, 99, 2.1, 6
    g = gauge_field(X, Y, 7)
    gw = [gauge_words(g, c) for c in (0, 1)]
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, layout=layout) as s:
        s.init().sweep(n)
        want = [s.read(c) ^ gw[c] for c in (ig.BLACK, ig.WHITE)]
    with ig.IsingSlab(X, Y, seed=seed, temp=temp, J_prob=0.5, layout=layout) as s:
        s.init()
        s.write_couplings(ig.WHITE, mattis_nibbles(g, 0))
        s.write_couplings(ig.BLACK, mattis_nibbles(g, 1))
        for c in (ig.BLACK, ig.WHITE):
            s.write(c, s.read(c) ^ gw[c])
        s.sweep(n)
        for c in (ig.BLACK, ig.WHITE):
            assert np.array_equal(s.read(c), want[c]), c


@ALL_LAYOUTS
def test_swapped_couplings_are_the_symmetric_bond_model(gpu, layout):
    """What each site applies with the arrays as -J draws them (the black update reads hamW, the white one hamB, optimized/main.cu:1774,
    :1795) is NOT a symmetric set of bonds; after ising_swap_couplings (cuIsing --J-symmetric) it is: J_ij = J_ji on every bond."""
    from _gauge import bonds_symmetric, site_nibbles
    X, Y = 8192, 64
    with ig.IsingSlab(X, Y, seed=31, temp=1.5, J_prob=0.4, layout=layout) as s:
        s.init().init_couplings()
        hb, hw = s.read_couplings(ig.BLACK), s.read_couplings(ig.WHITE)
        assert bonds_symmetric(site_nibbles(hb, hw))      # hamB = the black sites' bonds, hamW = the white sites': a consistent pair ...
        assert not bonds_symmetric(site_nibbles(hw, hb))  # ... but not the way the updates read them
        s.swap_couplings()
        assert np.array_equal(s.read_couplings(ig.BLACK), hw) and np.array_equal(s.read_couplings(ig.WHITE), hb)
        # now the black update reads array WHITE = hb = the black sites' bonds, the white update array BLACK = hw
        assert bonds_symmetric(site_nibbles(s.read_couplings(ig.WHITE), s.read_couplings(ig.BLACK)))
        s.sweep(3)  # (and the update runs on them: the energy moves towards the ground state of the frustrated model at low T)
        assert s.count()[0] + s.count()[1] == X * Y


def test_cli_symmetric_bonds_flag(gpu):
    base = ["-x", "4096", "-y", "256", "-n", "8", "-p", "4", "-t", "1.2", "-s", "5", "-J", "0.3"]
    ref = subprocess.run([CLI] + base, capture_output=True, text=True, timeout=300)
    sym = subprocess.run([CLI] + base + ["--J-symmetric"], capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0 and sym.returncode == 0, ref.stderr + sym.stderr
    assert "symmetric bonds" in sym.stdout and "symmetric bonds" not in ref.stdout
    pick = lambda out: re.findall(r"magnetization: +[\d.]+, up_s: +(\d+)", out)
    assert len(pick(ref.stdout)) == len(pick(sym.stdout)) >= 3
    assert pick(ref.stdout)[0] == pick(sym.stdout)[0] and pick(ref.stdout)[-1] != pick(sym.stdout)[-1]  # same start, another model
    # ring of two slabs on one device == one slab, also with the arrays swapped
    two = subprocess.run([CLI, "-x", "4096", "-y", "128", "-d", "2", "--devmap", "0,0"] + base[4:] + ["--J-symmetric"], capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stderr
    assert pick(two.stdout) == pick(sym.stdout)


def test_swapped_couplings_in_a_ring_with_ghost_rows(gpu):
    """Ballot ring slabs keep ghost rows of the coupling arrays too (the fused launches update ghost rows): swapped on every slab,
    three slabs through the C-ABI ring give the single slab's spins."""
    X, Y, n = 16384, 1536, 3
    with ig.IsingSlab(X, Y, seed=8, temp=1.3, J_prob=0.3, layout=ig.LAYOUT_BALLOT) as one:
        one.init().init_couplings().swap_couplings().sweep(40)
        ref = (one.read(ig.BLACK), one.read(ig.WHITE))
    slabs = [ig.IsingSlab(X, Y // n, seed=8, temp=1.3, nslabs=n, slab=k, J_prob=0.3, layout=ig.LAYOUT_BALLOT) for k in range(n)]
    try:
        ring = ig.SlabSet(slabs).init()
        assert slabs[0].ghost_ptrs(ig.BLACK)[0] > 1
        for s in slabs:
            s.swap_couplings()
        ring.sweep(33).sweep(7)
        ring.synchronize()
        assert np.array_equal(np.concatenate([s.read(ig.BLACK) for s in slabs]), ref[0])
        assert np.array_equal(np.concatenate([s.read(ig.WHITE) for s in slabs]), ref[1])
    finally:
        for s in slabs:
            s.close()
